"""Coefficients of the f32 polynomial for 2^f, f in [-1/2, 1/2], used by the `tol` math mode (cell_update.h: split_potential).

P(f) = 1 + f (c1 + f (c2 + ... + f c_n)), every c_k an f32 number.  Fitted by least squares on Chebyshev nodes in the
relative error (P - 2^f) / 2^f (within a small factor of minimax), with the coefficients rounded to f32 one after the
other, lowest first, and the rest refitted after each rounding so that later coefficients absorb the rounding of
earlier ones.  Prints the coefficients as C hex floats and the error of the ROUNDED polynomial in exact arithmetic:
max |rel err| and its mean over uniform f (the bias; must be << 1e-3 ulp = 6e-11).

    python tools/gen_exp2_poly.py 7
"""
import struct
import sys

import mpmath as mp

mp.mp.prec = 200


def f32(x):
    return mp.mpf(struct.unpack("f", struct.pack("f", float(x)))[0])


def fit(degree, fixed):
    """LSQ for c_{len(fixed)+1..degree} given the fixed leading ones (c_1..), relative-error weighted."""
    nodes = [mp.cos(mp.pi * (2 * i + 1) / (2 * 400)) / 2 for i in range(400)]
    k0 = len(fixed) + 1
    rows, rhs = [], []
    for f in nodes:
        w = mp.power(2, -f)
        known = 1 + sum(c * f ** (i + 1) for i, c in enumerate(fixed))
        rows.append([w * f ** k for k in range(k0, degree + 1)])
        rhs.append(w * (mp.power(2, f) - known))
    A = mp.matrix(rows)
    b = mp.matrix(rhs)
    return list(mp.lu_solve(A.T * A, A.T * b))


def main():
    degree = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    fixed = []
    while len(fixed) < degree:
        c = fit(degree, fixed)
        fixed.append(f32(c[0]))
    worst, mean, n = mp.mpf(0), mp.mpf(0), 20001
    for i in range(n):
        f = mp.mpf(i) / (n - 1) - mp.mpf(1) / 2
        p = 1 + sum(c * f ** (k + 1) for k, c in enumerate(fixed))
        e = p / mp.power(2, f) - 1
        worst = max(worst, abs(e))
        mean += e / n
    print("degree %d: max rel err %.3e (%.4f ulp of 2^-24), mean %.3e" % (degree, worst, worst * 2 ** 24, mean))
    for k, c in enumerate(fixed):
        print("  c%d = %s  /* %.10g */" % (k + 1, float(c).hex(), float(c)))


if __name__ == "__main__":
    main()
