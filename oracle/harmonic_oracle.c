/*
 * harmonic_oracle.c -- CPU ORACLE for the log-space harmonic relaxation path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  Nothing under epic_amd/ links,
 * imports or calls it; the shipped libepic.so has its own sources under epic_amd/csrc/.
 *
 * It restates, in plain C, the algorithm of the reference's CPU solver
 * (/root/reference/libepic/src/harmonic/harmonic_cpu.cpp) so that a checker exists on
 * the GPU box, where /root/reference does not.  Every function cites the reference
 * lines it follows.  Parity status: PINNED -- tests/test_oracle.py compares it bit for bit
 * with the reference sources compiled here into oracle/_ref/ (see Makefile) and with the
 * committed vectors in tests/golden/ that were produced by that reference build
 * (tests/golden/generate_goldens.py).
 *
 * Build: gcc -O3 -std=c11 -shared -fPIC (NO -ffast-math, NO -march=native: the
 * reference is built with plain -O3, libepic/Makefile:1-21, and its rounding sequence
 * is what parity is defined on).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* Same layout as the reference's struct (libepic/include/epic/harmonic/harmonic.h:44-64):
 * 80 bytes on x86-64 SysV. */
typedef struct OracleHarmonic {
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
} OracleHarmonic;

#define ORACLE_SUCCESS 0
#define ORACLE_SUCCESS_AND_CONVERGED 1
#define ORACLE_ERROR_INVALID_DATA 2

static inline float fmax2(float a, float b) { return a < b ? b : a; } /* std::max(a,b) */

/* ---- the per-cell update --------------------------------------------------------
 * harmonic_cpu.cpp:60-70 (2-D) and :110-123 (3-D).  Rounding sequence, spelled out:
 *   mx = max over neighbours                          (float)
 *   s  = ((expf(a-mx) + expf(b-mx)) + expf(c-mx)) + expf(d-mx) [+ ...]   (float, left-assoc)
 *   t  = mx + logf(s)                                 (float)
 *   u  = (float)((double)t - log(2.0 * n))            (double subtract, then round)
 */
static inline float cell_update_2d(float up, float down, float left, float right)
{
    float mx = fmax2(up, down);
    mx = fmax2(mx, left);
    mx = fmax2(mx, right);
    float s = expf(up - mx) + expf(down - mx) + expf(left - mx) + expf(right - mx);
    float t = mx + logf(s);
    return (float)((double)t - log(2.0 * 2));
}

static inline float cell_update_3d(float a0, float a1, float b0, float b1, float c0, float c1)
{
    float mx = fmax2(a0, a1);
    mx = fmax2(mx, b0);
    mx = fmax2(mx, b1);
    mx = fmax2(mx, c0);
    mx = fmax2(mx, c1);
    float s = expf(a0 - mx) + expf(a1 - mx) + expf(b0 - mx) + expf(b1 - mx) + expf(c0 - mx) + expf(c1 - mx);
    float t = mx + logf(s);
    return (float)((double)t - log(2.0 * 3));
}

/* ---- red-black half-sweeps (the reference's scheme) ------------------------------ */

/* harmonic_cpu.cpp:38-78.  Returns the number of cells recomputed. */
static uint64_t rb_update_2d(OracleHarmonic *h, int check)
{
    const unsigned int m0 = h->m[0], m1 = h->m[1];
    uint64_t updates = 0;
    if (check) h->delta = 0.0f;
    for (unsigned int x0 = 1; x0 + 1 < m0; x0++) {
        unsigned int offset = (unsigned int)((h->currentIteration % 2) != (x0 % 2));
        for (unsigned int x1 = 1 + offset; x1 + 1 < m1; x1 += 2) {
            size_t c = (size_t)x0 * m1 + x1;
            if (h->locked[c]) continue;
            float prev = h->u[c];
            float v = cell_update_2d(h->u[c - m1], h->u[c + m1], h->u[c - 1], h->u[c + 1]);
            h->u[c] = v;
            updates++;
            if (check) h->delta = fmax2(h->delta, (float)fabs(prev - v));
        }
    }
    return updates;
}

/* The same 2-D half-sweep with its rows dealt to OpenMP threads.  NOT in the reference (which is single-threaded): within
 * a half-sweep every updated cell reads only cells of the other colour, so rows are independent and the result is the
 * sequential one bit for bit.  bench.py reports it as the build's own all-cores CPU figure, next to the reference's. */
int oracle_update_parallel_2d(OracleHarmonic *h, int threads)
{
    if (h == NULL || h->n != 2 || h->m == NULL || h->u == NULL || h->locked == NULL || threads < 1)
        return ORACLE_ERROR_INVALID_DATA;
    const unsigned int m0 = h->m[0], m1 = h->m[1];
    const unsigned int it = h->currentIteration;
    float *u = h->u;
    const unsigned int *locked = h->locked;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long long x0 = 1; x0 < (long long)m0 - 1; x0++) {
        unsigned int offset = (unsigned int)((it % 2) != ((unsigned int)x0 % 2));
        for (unsigned int x1 = 1 + offset; x1 + 1 < m1; x1 += 2) {
            size_t c = (size_t)x0 * m1 + x1;
            if (locked[c]) continue;
            u[c] = cell_update_2d(u[c - m1], u[c + m1], u[c - 1], u[c + 1]);
        }
    }
    h->currentIteration++;
    return ORACLE_SUCCESS;
}

/* ... and the check form (harmonic_cpu.cpp:203-220): max |du| over the cells the half-sweep recomputes.  A maximum does not depend on the
 * order it is taken in, so delta is the sequential one bit for bit as well. */
int oracle_update_and_check_parallel_2d(OracleHarmonic *h, int threads)
{
    if (h == NULL || h->n != 2 || h->m == NULL || h->u == NULL || h->locked == NULL || threads < 1)
        return ORACLE_ERROR_INVALID_DATA;
    const unsigned int m0 = h->m[0], m1 = h->m[1];
    const unsigned int it = h->currentIteration;
    float *u = h->u;
    const unsigned int *locked = h->locked;
    float delta = 0.0f;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(max : delta)
    for (long long x0 = 1; x0 < (long long)m0 - 1; x0++) {
        unsigned int offset = (unsigned int)((it % 2) != ((unsigned int)x0 % 2));
        for (unsigned int x1 = 1 + offset; x1 + 1 < m1; x1 += 2) {
            size_t c = (size_t)x0 * m1 + x1;
            if (locked[c]) continue;
            const float prev = u[c];
            const float v = cell_update_2d(u[c - m1], u[c + m1], u[c - 1], u[c + 1]);
            u[c] = v;
            delta = fmax2(delta, (float)fabs(prev - v));
        }
    }
    h->delta = delta;
    h->currentIteration++;
    return (h->delta < h->epsilon) ? ORACLE_SUCCESS_AND_CONVERGED : ORACLE_SUCCESS;
}

/* harmonic_complete_cpu's loop (harmonic_cpu.cpp:136-184; oracle_complete below) with the half-sweeps dealt to threads: the reference's field,
 * iteration count and delta bit for bit, in 1 / threads of the time -- what makes a CPU-side statement of the reference's converged field
 * possible at the benchmark's own size (8192^2: 45 001 half-sweeps, ~15 h on one core; tests/golden/generate_8192_golden.py).  progress_every > 0:
 * a line on stderr every so many iterations. */
int oracle_complete_parallel_2d(OracleHarmonic *h, int threads, unsigned int progress_every)
{
    if (h == NULL || h->n != 2 || h->m == NULL || h->u == NULL || h->locked == NULL || h->epsilon <= 0.0 || threads < 1 ||
        h->numIterationsToStaggerCheck == 0)
        return ORACLE_ERROR_INVALID_DATA;
    unsigned int mMax = h->m[0] > h->m[1] ? h->m[0] : h->m[1];
    h->currentIteration = 0;
    h->delta = h->epsilon + 1.0;
    int result = ORACLE_SUCCESS;
    while (result != ORACLE_SUCCESS_AND_CONVERGED || h->currentIteration < mMax) {
        if (h->currentIteration % h->numIterationsToStaggerCheck == 0) {
            result = oracle_update_and_check_parallel_2d(h, threads);
            if (progress_every && (h->currentIteration - 1) % progress_every == 0)
                fprintf(stderr, "[oracle_complete_parallel_2d] iteration %u delta %.6e\n", h->currentIteration, (double)h->delta);
        } else {
            result = oracle_update_parallel_2d(h, threads);
        }
        if (result != ORACLE_SUCCESS && result != ORACLE_SUCCESS_AND_CONVERGED) return result;
    }
    return ORACLE_SUCCESS;
}

/* harmonic_cpu.cpp:81-133. */
static uint64_t rb_update_3d(OracleHarmonic *h, int check)
{
    const unsigned int m0 = h->m[0], m1 = h->m[1], m2 = h->m[2];
    const size_t s0 = (size_t)m1 * m2, s1 = m2;
    uint64_t updates = 0;
    if (check) h->delta = 0.0f;
    for (unsigned int x0 = 1; x0 + 1 < m0; x0++) {
        for (unsigned int x1 = 1; x1 + 1 < m1; x1++) {
            unsigned int offset = (unsigned int)((h->currentIteration % 2) != (x0 % 2));
            if (x1 % 2 == 0) offset = !offset;
            for (unsigned int x2 = 1 + offset; x2 + 1 < m2; x2 += 2) {
                size_t c = x0 * s0 + x1 * s1 + x2;
                if (h->locked[c]) continue;
                float prev = h->u[c];
                float v = cell_update_3d(h->u[c - s0], h->u[c + s0], h->u[c - s1], h->u[c + s1],
                                         h->u[c - 1], h->u[c + 1]);
                h->u[c] = v;
                updates++;
                if (check) h->delta = fmax2(h->delta, (float)fabs(prev - v));
            }
        }
    }
    return updates;
}

/* The 3-D half-sweep (harmonic_cpu.cpp:81-133, rb_update_3d above) with its planes dealt to OpenMP threads: every updated cell reads only cells of
 * the other colour, so planes are independent and field and max |du| are the sequential ones bit for bit.  check: also h->delta. */
static int rb_update_3d_parallel(OracleHarmonic *h, int check, int threads)
{
    const unsigned int m0 = h->m[0], m1 = h->m[1], m2 = h->m[2];
    const size_t s0 = (size_t)m1 * m2, s1 = m2;
    const unsigned int it = h->currentIteration;
    float *u = h->u;
    const unsigned int *locked = h->locked;
    float delta = 0.0f;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(max : delta)
    for (long long x0 = 1; x0 < (long long)m0 - 1; x0++) {
        for (unsigned int x1 = 1; x1 + 1 < m1; x1++) {
            unsigned int offset = (unsigned int)((it % 2) != ((unsigned int)x0 % 2));
            if (x1 % 2 == 0) offset = !offset;
            for (unsigned int x2 = 1 + offset; x2 + 1 < m2; x2 += 2) {
                size_t c = (size_t)x0 * s0 + x1 * s1 + x2;
                if (locked[c]) continue;
                const float prev = u[c];
                const float v = cell_update_3d(u[c - s0], u[c + s0], u[c - s1], u[c + s1], u[c - 1], u[c + 1]);
                u[c] = v;
                if (check) delta = fmax2(delta, (float)fabs(prev - v));
            }
        }
    }
    if (check) h->delta = delta;
    h->currentIteration++;
    return ORACLE_SUCCESS;
}

/* harmonic_complete_cpu's loop for n = 3 with the half-sweeps dealt to threads (as oracle_complete_parallel_2d): the reference's converged 512^3
 * field of BASELINE configs[4] on the CPU (tests/golden/generate_8192_golden.py --cube 512). */
int oracle_complete_parallel_3d(OracleHarmonic *h, int threads, unsigned int progress_every)
{
    if (h == NULL || h->n != 3 || h->m == NULL || h->u == NULL || h->locked == NULL || h->epsilon <= 0.0 || threads < 1 ||
        h->numIterationsToStaggerCheck == 0)
        return ORACLE_ERROR_INVALID_DATA;
    unsigned int mMax = 0;
    for (unsigned int i = 0; i < 3; i++) mMax = h->m[i] > mMax ? h->m[i] : mMax;
    h->currentIteration = 0;
    h->delta = h->epsilon + 1.0;
    int converged = 0;
    while (!converged || h->currentIteration < mMax) {
        const int check = (h->currentIteration % h->numIterationsToStaggerCheck) == 0;
        rb_update_3d_parallel(h, check, threads);
        converged = check && h->delta < h->epsilon;
        if (check && progress_every && (h->currentIteration - 1) % progress_every == 0)
            fprintf(stderr, "[oracle_complete_parallel_3d] iteration %u delta %.6e\n", h->currentIteration, (double)h->delta);
    }
    return ORACLE_SUCCESS;
}

static uint64_t g_updates; /* cells recomputed since oracle_reset_counters() */

void oracle_reset_counters(void) { g_updates = 0; }
uint64_t oracle_cell_updates(void) { return g_updates; }

/* harmonic_cpu.cpp:187-200 (n == 4 is a no-op that still counts an iteration). */
int oracle_update(OracleHarmonic *h)
{
    if (h->n == 2) g_updates += rb_update_2d(h, 0);
    else if (h->n == 3) g_updates += rb_update_3d(h, 0);
    h->currentIteration++;
    return ORACLE_SUCCESS;
}

/* harmonic_cpu.cpp:203-220. */
int oracle_update_and_check(OracleHarmonic *h)
{
    if (h->n == 2) g_updates += rb_update_2d(h, 1);
    else if (h->n == 3) g_updates += rb_update_3d(h, 1);
    h->currentIteration++;
    return (h->delta < h->epsilon) ? ORACLE_SUCCESS_AND_CONVERGED : ORACLE_SUCCESS;
}

/* harmonic_cpu.cpp:136-184.  Exit only right after a check sweep that converged AND
 * with currentIteration >= max(m[i]). */
int oracle_complete(OracleHarmonic *h)
{
    if (h == NULL || h->m == NULL || h->u == NULL || h->locked == NULL || h->epsilon <= 0.0)
        return ORACLE_ERROR_INVALID_DATA;
    unsigned int mMax = 0;
    for (unsigned int i = 0; i < h->n; i++) mMax = h->m[i] > mMax ? h->m[i] : mMax;
    h->currentIteration = 0;
    h->delta = h->epsilon + 1.0;
    int result = ORACLE_SUCCESS;
    while (result != ORACLE_SUCCESS_AND_CONVERGED || h->currentIteration < mMax) {
        if (h->currentIteration % h->numIterationsToStaggerCheck == 0)
            result = oracle_update_and_check(h);
        else
            result = oracle_update(h);
    }
    return ORACLE_SUCCESS;
}

/* ---- Jacobi sweeps: the SAME per-cell update applied to all unlocked interior cells
 * from the previous sweep's values (what the HIP kernels do; SURVEY.md §7 shows the
 * scheme does not move the f32 stagnation point).  Not in the reference: it exists so
 * that fixed-sweep-count GPU results can be compared cell by cell. ------------------ */

/* (rows are dealt to OpenMP threads on large grids -- a Jacobi sweep reads `in` only, so the result is the sequential one
 * bit for bit; the full-size whole-field tests sweep 8192^2 and 512^3 with it) */
#define ORACLE_OMP_MIN_CELLS ((size_t)1 << 17)
static void jacobi_sweep_2d(const OracleHarmonic *h, const float *in, float *out, float *delta)
{
    const unsigned int m0 = h->m[0], m1 = h->m[1];
    float d = 0.0f;
    uint64_t n = 0;
    memcpy(out, in, (size_t)m0 * m1 * sizeof(float));
#pragma omp parallel for schedule(static) reduction(max : d) reduction(+ : n) if ((size_t)m0 * m1 >= ORACLE_OMP_MIN_CELLS)
    for (unsigned int x0 = 1; x0 < m0 - 1; x0++) {
        for (unsigned int x1 = 1; x1 + 1 < m1; x1++) {
            size_t c = (size_t)x0 * m1 + x1;
            if (h->locked[c]) continue;
            float v = cell_update_2d(in[c - m1], in[c + m1], in[c - 1], in[c + 1]);
            out[c] = v;
            d = fmax2(d, (float)fabs(in[c] - v));
            n++;
        }
    }
    g_updates += n;
    if (delta) *delta = d;
}

static void jacobi_sweep_3d(const OracleHarmonic *h, const float *in, float *out, float *delta)
{
    const unsigned int m0 = h->m[0], m1 = h->m[1], m2 = h->m[2];
    const size_t s0 = (size_t)m1 * m2, s1 = m2;
    float d = 0.0f;
    uint64_t n = 0;
    memcpy(out, in, (size_t)m0 * s0 * sizeof(float));
#pragma omp parallel for schedule(static) reduction(max : d) reduction(+ : n) if ((size_t)m0 * s0 >= ORACLE_OMP_MIN_CELLS)
    for (unsigned int x0 = 1; x0 < m0 - 1; x0++)
        for (unsigned int x1 = 1; x1 + 1 < m1; x1++)
            for (unsigned int x2 = 1; x2 + 1 < m2; x2++) {
                size_t c = x0 * s0 + x1 * s1 + x2;
                if (h->locked[c]) continue;
                float v = cell_update_3d(in[c - s0], in[c + s0], in[c - s1], in[c + s1], in[c - 1], in[c + 1]);
                out[c] = v;
                d = fmax2(d, (float)fabs(in[c] - v));
                n++;
            }
    g_updates += n;
    if (delta) *delta = d;
}

static size_t num_cells(const OracleHarmonic *h)
{
    size_t c = 1;
    for (unsigned int i = 0; i < h->n; i++) c *= h->m[i];
    return c;
}

/* Run `sweeps` Jacobi sweeps on h->u in place; h->delta = max |du| of the LAST sweep;
 * currentIteration advances by `sweeps`. */
int oracle_jacobi_run(OracleHarmonic *h, unsigned int sweeps)
{
    if (h == NULL || h->m == NULL || h->u == NULL || h->locked == NULL || (h->n != 2 && h->n != 3))
        return ORACLE_ERROR_INVALID_DATA;
    size_t cells = num_cells(h);
    float *a = h->u, *b = (float *)malloc(cells * sizeof(float));
    if (!b) return ORACLE_ERROR_INVALID_DATA;
    for (unsigned int s = 0; s < sweeps; s++) {
        if (h->n == 2) jacobi_sweep_2d(h, a, b, &h->delta);
        else jacobi_sweep_3d(h, a, b, &h->delta);
        float *t = a; a = b; b = t;
        h->currentIteration++;
    }
    if (a != h->u) { memcpy(h->u, a, cells * sizeof(float)); free(a); }
    else free(b);
    return ORACLE_SUCCESS;
}

/* Jacobi driven by the reference's loop rule (harmonic_cpu.cpp:158-173): delta is
 * looked at only on sweeps with currentIteration % stagger == 0 (before the increment).
 *
 * Handover (the library's harmonic_execute_gpu does the same, epic_amd/csrc/driver_loop.hip): a Jacobi iteration is two
 * interleaved red-black chains (the cells of one colour at even iterations and of the other at odd ones never meet the
 * rest), and in f32 the two may stagnate a unit in the last place apart -- then every cell flips between them for ever
 * and max |du| never falls below eps (first seen on the nav_core plugin's second makePlan, tests/test_gpu_plugin_replay.py).
 * So: at the first check with delta < 1 that is not below the previous check's delta, the iteration continues as the
 * reference's red-black half-sweeps in place, which stop as the reference stops.  ORACLE_JACOBI_HANDOVER_DELTA is that 1:
 * far below the ~1e6 of a front still moving, far above any f32 flicker of a field the solver can hold. */
#define ORACLE_JACOBI_HANDOVER_DELTA 1.0f
/* The library's opt-in EPIC_HIP_JACOBI_CHECKS=reference (epic_amd/csrc/driver_loop.hip: run_block), stated for the checkers' Jacobi loops
 * (here and oracle_tol_complete): every CHECK iteration is the reference's red-black half-sweep of that iteration's colour
 * (harmonic_cpu.cpp:38-133), in place in the current Jacobi state.  After the Jacobi sweeps 1 .. k-1 the other colour holds the values
 * the reference's half-sweep k-1 left, so the state after the check is the reference's after k half-sweeps, and delta the reference's. */
int g_oracle_jacobi_ref_checks = 0;
void oracle_set_jacobi_ref_checks(int on) { g_oracle_jacobi_ref_checks = on != 0; }
int oracle_jacobi_complete(OracleHarmonic *h)
{
    if (h == NULL || h->m == NULL || h->u == NULL || h->locked == NULL || h->epsilon <= 0.0 ||
        (h->n != 2 && h->n != 3) || h->numIterationsToStaggerCheck == 0)
        return ORACLE_ERROR_INVALID_DATA;
    size_t cells = num_cells(h);
    unsigned int mMax = 0;
    for (unsigned int i = 0; i < h->n; i++) mMax = h->m[i] > mMax ? h->m[i] : mMax;
    float *a = h->u, *b = (float *)malloc(cells * sizeof(float));
    if (!b) return ORACLE_ERROR_INVALID_DATA;
    h->currentIteration = 0;
    h->delta = h->epsilon + 1.0;
    int converged = 0, handed_over = 0;
    float last_check = -1.0f;   /* no check yet */
    while (!converged || h->currentIteration < mMax) {
        int check = (h->currentIteration % h->numIterationsToStaggerCheck) == 0;
        if (handed_over) {
            if (h->n == 2) g_updates += rb_update_2d(h, check);
            else g_updates += rb_update_3d(h, check);
            h->currentIteration++;
            converged = check && h->delta < h->epsilon;
            continue;
        }
        float d;
        if (check && g_oracle_jacobi_ref_checks) {
            float *own = h->u;
            h->u = a;   /* the half-sweep of this iteration's colour, in place in the current state */
            if (h->n == 2) g_updates += rb_update_2d(h, 1);
            else g_updates += rb_update_3d(h, 1);
            h->u = own;
            d = (float)h->delta;
        } else {
            if (h->n == 2) jacobi_sweep_2d(h, a, b, &d);
            else jacobi_sweep_3d(h, a, b, &d);
            float *t = a; a = b; b = t;
        }
        h->currentIteration++;
        if (check) {
            h->delta = d;
            converged = d < h->epsilon;
            if (!converged && d < ORACLE_JACOBI_HANDOVER_DELTA && last_check >= 0.0f && d >= last_check) {
                if (a != h->u) { memcpy(h->u, a, cells * sizeof(float)); free(a); a = h->u; b = NULL; }
                else { free(b); b = NULL; }
                handed_over = 1;
            }
            last_check = d;
        } else converged = 0;
    }
    if (!handed_over) {
        if (a != h->u) { memcpy(h->u, a, cells * sizeof(float)); free(a); }
        else free(b);
    }
    return ORACLE_SUCCESS;
}

/* ---- sparse cell edits: harmonic_utilities_cpu.cpp:38-76 ------------------------- */
int oracle_set_cells_2d(OracleHarmonic *h, unsigned int k, const unsigned int *v, const unsigned int *types)
{
    if (h == NULL || h->n == 0 || h->m == NULL || h->u == NULL || h->locked == NULL || k == 0 ||
        v == NULL || types == NULL)
        return ORACLE_ERROR_INVALID_DATA;
    for (unsigned int i = 0; i < k; i++) {
        unsigned int x = v[2 * i], y = v[2 * i + 1];
        if (y >= h->m[0] || x >= h->m[1]) continue;
        size_t c = (size_t)y * h->m[1] + x;
        if (types[i] == 0) { h->u[c] = 0.0f; h->locked[c] = 1; }
        else if (types[i] == 1) { h->u[c] = -1e6f; h->locked[c] = 1; }
        else if (types[i] == 2) { h->u[c] = -1e6f; h->locked[c] = 0; }
    }
    return ORACLE_SUCCESS;
}

/* ---- synthetic occupancy grids (not in the reference) ----
 * Counter-based hash so that C, numpy and the device generate identical maps:
 *   h = splitmix64_mix(seed ^ (idx * 0x9E3779B97F4A7C15)); obstacle iff (h >> 11) * 2^-53 < density
 * border cells are obstacles, the centre cell is the single goal (u = 0), the rest free (u = -1e6).
 */
static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void oracle_synthetic(unsigned int n, const unsigned int *m, uint64_t seed, double density, float *u,
                      unsigned int *locked)
{
    size_t cells = 1, goal = 0;
    for (unsigned int i = 0; i < n; i++) cells *= m[i];
    for (unsigned int i = 0; i < n; i++) goal = goal * m[i] + m[i] / 2;
    const uint64_t thresh = (uint64_t)(density * 9007199254740992.0); /* density * 2^53 */
#pragma omp parallel for schedule(static) if (cells >= ((size_t)1 << 20))
    for (size_t idx = 0; idx < cells; idx++) {
        size_t r = idx;
        int border = 0;
        for (int i = (int)n - 1; i >= 0; i--) {
            unsigned int x = (unsigned int)(r % m[i]);
            r /= m[i];
            if (x == 0 || x == m[i] - 1) border = 1;
        }
        uint64_t hsh = mix64(seed ^ ((uint64_t)idx * 0x9E3779B97F4A7C15ULL));
        int obstacle = border || ((hsh >> 11) < thresh);
        u[idx] = -1e6f;
        locked[idx] = (unsigned int)obstacle;
    }
    u[goal] = 0.0f;
    locked[goal] = 1;
}

/* A seeded NON-UNIFORM start for the full-size whole-field tests: every unlocked cell gets lo + (hi - lo) * hash01(seed, idx)
 * (the same counter-based hash), locked cells keep what they hold.  From the uniform start (-1e6 everywhere but the goal) a
 * store that lands in the wrong row or strip far from the goal writes the seed over the seed and cannot be seen. */
void oracle_scramble_free(unsigned int n, const unsigned int *m, uint64_t seed, float lo, float hi, float *u,
                          const unsigned int *locked)
{
    size_t cells = 1;
    for (unsigned int i = 0; i < n; i++) cells *= m[i];
#pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < cells; idx++) {
        if (locked[idx]) continue;
        const uint64_t hsh = mix64(seed ^ ((uint64_t)idx * 0x9E3779B97F4A7C15ULL));
        u[idx] = lo + (hi - lo) * (float)((double)(hsh >> 11) * 0x1p-53);
    }
}

/* ---- row-range Jacobi on a pitched local array (checker side of the slab-decomposition tests) ----------------
 * Sweeps rows [row_begin, row_end) of a rows x pitch array (first `cols` columns meaningful) from `in` to `out`
 * with the same per-cell update; cells with locked != 0, the first/last column and rows 0 / rows-1 are copied.
 * Returns max |du| over the rows swept.  Not in the reference (it has no multi-GPU code). */
float oracle_jacobi_rows_2d(const float *in, float *out, const unsigned int *locked, unsigned int rows,
                            unsigned int cols, unsigned int pitch, unsigned int row_begin, unsigned int row_end)
{
    float d = 0.0f;
    for (unsigned int r = row_begin; r < row_end && r < rows; r++) {
        for (unsigned int c = 0; c < pitch; c++) {
            size_t i = (size_t)r * pitch + c;
            int fixed = c >= cols || c == 0 || c == cols - 1 || r == 0 || r == rows - 1 || locked[(size_t)r * cols + c];
            if (fixed) { out[i] = in[i]; continue; }
            float v = cell_update_2d(in[i - pitch], in[i + pitch], in[i - 1], in[i + 1]);
            out[i] = v;
            d = fmax2(d, (float)fabs(in[i] - v));
        }
    }
    return d;
}

/* The red-black form of the same: one in-place half-sweep of rows [row_begin, row_end) of a rows x pitch array, updating
 * the unlocked interior cells with (r + c + parity) odd -- harmonic_cpu.cpp:46-51 with parity standing for
 * currentIteration (a slab whose local row 0 is global row `top` passes currentIteration + top).  Returns max |du|. */
float oracle_redblack_rows_2d(float *u, const unsigned int *locked, unsigned int rows, unsigned int cols,
                              unsigned int pitch, unsigned int row_begin, unsigned int row_end, unsigned int parity)
{
    float d = 0.0f;
    for (unsigned int r = row_begin; r < row_end && r < rows; r++) {
        if (r == 0 || r == rows - 1) continue;
        for (unsigned int c = 1; c + 1 < cols; c++) {
            if (((r + c + parity) & 1u) == 0u || locked[(size_t)r * cols + c]) continue;
            size_t i = (size_t)r * pitch + c;
            float v = cell_update_2d(u[i - pitch], u[i + pitch], u[i - 1], u[i + 1]);
            d = fmax2(d, (float)fabs(u[i] - v));
            u[i] = v;
        }
    }
    return d;
}

/* ---- the libm under the reference's arithmetic, array form ------------------------------------------------------
 * harmonic_cpu.cpp:65-70 calls std::exp / std::log on floats, i.e. this host's libm expf / logf.  The device restates
 * them (epic_amd/csrc/cell_update.h); this is the checker for that restatement over WHOLE input ranges: the inputs are
 * the n consecutive float bit patterns first_bits, first_bits + 1, ...; got[i] is the device's result for the i-th.
 * Returns the number of results whose bits differ from libm's; *first_bad = index of the first one (if any).
 * which = 0: expf, 1: logf. */
size_t oracle_libm_mismatches(int which, uint32_t first_bits, size_t n, const float *got, size_t *first_bad, int threads)
{
    size_t bad = 0, first = (size_t)-1;
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(+ : bad) reduction(min : first)
    for (size_t i = 0; i < n; i++) {
        union { uint32_t u; float f; } x, want, have;
        x.u = first_bits + (uint32_t)i;
        want.f = which ? logf(x.f) : expf(x.f);
        have.f = got[i];
        if (want.u != have.u) {
            bad++;
            if (i < first) first = i;
        }
    }
    if (first_bad) *first_bad = first;
    return bad;
}
