/* cr_study: the reference's red-black iteration and rounding sequence (harmonic_cpu.cpp:38-78, :136-184) with expf/logf replaced by
 * CORRECTLY ROUNDED exp/log (double libm, rounded once to float).  How far from the reference's converged field does the closest
 * non-bit-identical arithmetic land?  usage: cr_study m0 m1 u0 locked golden [variant]   variant 0 = cr exp + cr log, 1 = glibc expf + cr log, 2 = cr exp + glibc logf */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
int main(int argc, char **argv)
{
    unsigned m0 = atoi(argv[1]), m1 = atoi(argv[2]);
    size_t cells = (size_t)m0 * m1;
    float *u = malloc(cells * 4), *g = malloc(cells * 4);
    unsigned *lk = malloc(cells * 4);
    FILE *f = fopen(argv[3], "rb"); if (fread(u, 4, cells, f) != cells) return 2; fclose(f);
    f = fopen(argv[4], "rb"); if (fread(lk, 4, cells, f) != cells) return 2; fclose(f);
    f = fopen(argv[5], "rb"); if (fread(g, 4, cells, f) != cells) return 2; fclose(f);
    int variant = argc > 6 ? atoi(argv[6]) : 0;
    unsigned it = 0, mMax = m0 > m1 ? m0 : m1;
    float delta = 2.0f; int conv = 0;
    long diff_exp = 0, n_exp = 0;
    while (!conv || it < mMax) {
        int check = it % 100 == 0;
        float d = 0.0f;
        for (unsigned x0 = 1; x0 < m0 - 1; x0++) {
            unsigned off = (it % 2 != x0 % 2);
            for (unsigned x1 = 1 + off; x1 < m1 - 1; x1 += 2) {
                size_t c = (size_t)x0 * m1 + x1;
                if (lk[c]) continue;
                float a = u[c - m1], b = u[c + m1], l = u[c - 1], r = u[c + 1];
                float mx = fmaxf(fmaxf(fmaxf(a, b), l), r);
                float nb[4] = {a, b, l, r}, e[4];
                for (int k = 0; k < 4; k++) {
                    float dd = nb[k] - mx;
                    float cr = (float)exp((double)dd), gl = expf(dd);
                    if (it > 90000) { n_exp++; diff_exp += cr != gl; }
                    e[k] = (variant == 1) ? gl : cr;
                }
                float s = ((e[0] + e[1]) + e[2]) + e[3];
                float ls = (variant == 2) ? logf(s) : (float)log((double)s);
                float t = mx + ls;
                float v = (float)((double)t - log(4.0));
                if (check) d = fmaxf(d, fabsf(v - u[c]));
                u[c] = v;
            }
        }
        it++;
        if (check) { delta = d; conv = d < 1e-6f; } else conv = 0;
        if (it > 400000) break;
    }
    double worst = 0, wabs = 0, sum = 0; long n = 0, nbad = 0;
    for (size_t i = 0; i < cells; i++) {
        if (lk[i] || g[i] <= -9e5f) continue;
        double e = (double)u[i] - g[i], rel = fabs(e) / fmax(1.0, fabs((double)g[i]));
        if (rel > worst) worst = rel;
        if (fabs(e) > wabs) wabs = fabs(e);
        nbad += rel > 1e-5; sum += e; n++;
    }
    printf("variant %d: %u iterations, delta %.3e, max rel %.3e, max abs %.3e, mean signed %.3e, cells over 1e-5: %ld; cr exp != glibc expf on %.4f %% of late terms\n",
           variant, it, delta, worst, wabs, sum / n, nbad, n_exp ? 100.0 * diff_exp / n_exp : 0.0);
    return 0;
}
