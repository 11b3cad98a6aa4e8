/* df32_study.c -- CPU emulation of a candidate f32 "double-float" exp/log (what a packed-f32 GPU kernel would run),
 * to answer two questions before any kernel is written:
 *   1. error statistics against libm / f64: max ulp, mean signed error (bias), mismatch rate with expf/logf;
 *   2. does a Jacobi relaxation that uses it converge to the reference's field within the 1e-5 parity bar on the
 *      ill-conditioned reference maps?  (usage: df32_study <m0> <m1> <u0.f32> <locked.u32> <golden.f32> [eps])
 * Build: gcc -O2 -ffp-contract=off -mfma tools/df32_study.c -o /tmp/df32_study -lm   (fmaf must be a real fma)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static float Th[32], Tl[32];              /* 2^(j/32) = Th + Tl */
static float LInv[128], LHi[128], LLo[128]; /* per (k, i): 1/c_i, ln(c_i) + k ln2 as hi + lo */
static float L1, L2;                      /* ln2/32 = L1 + L2, L1 with 11 significant bits */
#define OFF 0x3f330000u

static void init_tables(void)
{
    for (int j = 0; j < 32; j++) { double v = exp2(j / 32.0); Th[j] = (float)v; Tl[j] = (float)(v - (double)Th[j]); }
    double l = log(2.0) / 32.0;
    L1 = u2f(f2u((float)l) & 0xffffe000u);
    L2 = (float)(l - (double)L1);
    /* z in [OFF, 2 OFF) split into 32 sub-intervals of 2^-5 of the binade (mantissa bits 22..18 after the OFF shift) */
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < 32; i++) {
            /* centre of sub-interval i: bits = OFF + (i << 18) + (1 << 17) */
            float c = u2f(OFF + ((uint32_t)i << 18) + (1u << 17));
            float inv = (float)(1.0 / (double)c);
            double lc = -log((double)inv) + k * log(2.0);   /* ln(1/inv) exactly for the rounded inv */
            LInv[k * 32 + i] = inv;
            LHi[k * 32 + i] = (float)lc;
            LLo[k * 32 + i] = (float)(lc - (double)(float)lc);
        }
}

static inline float df_exp(float d)
{
    const float c = 46.16624130844683f, MAGIC = 12582912.0f;
    if (d < -104.0f) d = -104.0f;
    float zm = fmaf(d, c, MAGIC);
    float nf = zm - MAGIC;
    int32_t n = (int32_t)(f2u(zm) - f2u(MAGIC));
    float r = fmaf(nf, -L1, d);
    r = fmaf(nf, -L2, r);
    float q = fmaf(r, 1.0f / 24.0f, 1.0f / 6.0f);
    q = fmaf(r, q, 0.5f);
    float r2 = r * r;
    float p = fmaf(r2, q, r);
    int j = n & 31, k = n >> 5;
    float y = Th[j] + fmaf(Th[j], p, Tl[j]);
    return ldexpf(y, k);
}

static inline float df_log(float s)
{
    uint32_t ix = f2u(s), tmp = ix - OFF;
    uint32_t e = tmp >> 18;                    /* 32 k + i */
    float z = u2f(ix - (tmp & 0xff800000u));
    float r = fmaf(z, LInv[e], -1.0f);
    float q = fmaf(r, -0.25f, 1.0f / 3.0f);
    q = fmaf(r, q, -0.5f);
    float r2 = r * r;
    float t = fmaf(r2, q, LLo[e]);
    t = t + r;
    return LHi[e] + t;
}

static void stats(void)
{
    double sum = 0, mx = 0; uint64_t n = 0, mism = 0;
    for (uint32_t u = f2u(1.0f); u <= f2u(6.0f); u += 1) {
        float x = u2f(u), a = df_log(x), b = logf(x);
        double ref = log((double)x), ulp = (double)nextafterf(b, 10.0f) - (double)b;
        double e = ((double)a - ref) / ulp;
        sum += e; if (fabs(e) > mx) mx = fabs(e); n++; if (f2u(a) != f2u(b)) mism++;
    }
    printf("df_log [1,6]: n=%lu  mean err %+.5f ulp  max %.4f ulp  differs from logf in %.4f %%\n", n, sum / n, mx, 100.0 * mism / n);
    const double ranges[] = {0.01, 0.1, 1.0, 4.0, 20.0};
    for (int ri = 0; ri < 5; ri++) {
        sum = 0; mx = 0; n = 0; mism = 0;
        uint32_t hi = f2u((float)-ranges[ri]);
        for (uint32_t u = f2u(-1e-6f); u <= hi; u += 7) {
            float x = u2f(u), a = df_exp(x), b = expf(x);
            double ref = exp((double)x), ulp = (double)nextafterf(b, 10.0f) - (double)b;
            double e = ((double)a - ref) / ulp;
            sum += e; if (fabs(e) > mx) mx = fabs(e); n++; if (f2u(a) != f2u(b)) mism++;
        }
        printf("df_exp [-%g,0]: n=%lu  mean err %+.5f ulp  max %.4f ulp  differs from expf in %.4f %%\n", ranges[ri], n, sum / n, mx, 100.0 * mism / n);
    }
}

static inline float fmax2(float a, float b) { return a < b ? b : a; }
static inline float cell(float up, float dn, float lf, float rt)
{
    float mx = fmax2(fmax2(fmax2(up, dn), lf), rt);
    float s = df_exp(up - mx) + df_exp(dn - mx) + df_exp(lf - mx) + df_exp(rt - mx);
    float t = mx + df_log(s);
    return (float)((double)t - log(4.0));
}

int main(int argc, char **argv)
{
    init_tables();
    if (argc < 6) { stats(); return 0; }
    unsigned m0 = atoi(argv[1]), m1 = atoi(argv[2]);
    size_t cells = (size_t)m0 * m1;
    float *a = malloc(cells * 4), *b = malloc(cells * 4), *g = malloc(cells * 4);
    unsigned *lk = malloc(cells * 4);
    FILE *f = fopen(argv[3], "rb"); if (!f || fread(a, 4, cells, f) != cells) return 2; fclose(f);
    f = fopen(argv[4], "rb"); if (!f || fread(lk, 4, cells, f) != cells) return 2; fclose(f);
    f = fopen(argv[5], "rb"); if (!f || fread(g, 4, cells, f) != cells) return 2; fclose(f);
    float eps = argc > 6 ? (float)atof(argv[6]) : 1e-6f;
    unsigned stagger = 100, mMax = m0 > m1 ? m0 : m1, it = 0;
    int conv = 0;
    while (!conv || it < mMax) {
        int check = it % stagger == 0;
        float d = 0;
        memcpy(b, a, cells * 4);
        for (unsigned r = 1; r + 1 < m0; r++)
            for (unsigned c = 1; c + 1 < m1; c++) {
                size_t i = (size_t)r * m1 + c;
                if (lk[i]) continue;
                float v = cell(a[i - m1], a[i + m1], a[i - 1], a[i + 1]);
                b[i] = v;
                d = fmax2(d, fabsf(a[i] - v));
            }
        float *t = a; a = b; b = t;
        it++;
        conv = check ? d < eps : 0;
        if (it % 5000 == 0) { fprintf(stderr, "  sweep %u delta %.3e\n", it, d); }
        if (it > 400000) break;
    }
    double worst = 0, worst_abs = 0;
    for (size_t i = 0; i < cells; i++) {
        if (lk[i] || g[i] <= -9e5f) continue;
        double e = fabs((double)a[i] - g[i]);
        double rel = e / fmax(1.0, fabs((double)g[i]));
        if (rel > worst) worst = rel;
        if (e > worst_abs) worst_abs = e;
    }
    printf("df32 Jacobi: %u sweeps; vs reference golden: max rel %.3e, max abs %.3e\n", it, worst, worst_abs);
    return 0;
}
