#!/usr/bin/env python3
"""The local bias of the tol split -- a hypothesis about the tol iteration's offset on plateaus that was tested and refuted
(profiles/r03_experiments.txt item 12; DESIGN.md section 7: removing the bias left umass.png where it was).

The split e^u = q 2^n of the tol mode (cell_update.h: tol_split2; oracle/tol_checker.c) is unbiased over wide ranges of u
(mean relative error ~1e-11) but NOT locally: this script bins the error by the fraction f = u log2(e) - n and attributes it
to the stages of the split, emulated in extended precision -- the polynomial itself, its f32 Horner evaluation, and the two
roundings of the argument reduction  f1 = RN(u HI - n),  f = RN(u LO + f1).   python tools/split_bias_study.py"""
import numpy as np

rng = np.random.default_rng(13)
HI, LO = float.fromhex("0x1.715476p+0"), float.fromhex("0x1.4ae0bep-26")
L = np.longdouble("1.44269504088896340735992468100189214")
C = [float.fromhex(x) for x in ("0x1.62e43p-1", "0x1.ebfbep-3", "0x1.c6b072p-5", "0x1.3b2a4ap-7", "0x1.5da0f4p-10", "0x1.44227cp-13",
                                "0x1.e5ba06p-17")]
u = rng.uniform(float((-24 - 0.5) / L), float((-24 + 7.5) / L), 8_000_000).astype(np.float32)   # eight periods of f around u = -14
ul = u.astype(np.longdouble)
z = ul * L
n = np.rint(z)
f_true = z - n
x1 = ul * np.longdouble(HI) - n
f1 = x1.astype(np.float32)
x2 = ul * np.longdouble(LO) + f1.astype(np.longdouble)
f2 = x2.astype(np.float32)


def horner(f, single):
    p = np.full_like(f, C[6])
    for c in (C[5], C[4], C[3], C[2], C[1], C[0]):
        p = p * f + c
        if single:
            p = p.astype(np.float32).astype(np.float64)
    q = p * f + 1.0
    return q.astype(np.float32).astype(np.float64) if single else q


ref = np.exp2(np.asarray(f_true, dtype=np.float64))
bins = np.linspace(-0.5, 0.5, 9)
idx = np.digitize(np.asarray(f_true, dtype=np.float64), bins) - 1


def show(name, e):
    e = np.asarray(e, dtype=np.float64)
    print("%-52s mean %+.2e   per eighth of f: %s" % (name, e.mean(), " ".join("%+.1e" % e[idx == b].mean() for b in range(8))))


print("relative error of q against 2^f (1 ulp of q = 6e-8 .. 1.2e-7):")
show("polynomial alone (exact f, exact Horner)", horner(np.asarray(f_true, dtype=np.float64), False) / ref - 1)
show("exact f, f32 Horner", horner(np.asarray(f_true, dtype=np.float64).astype(np.float32).astype(np.float64), True) / ref - 1)
show("f32 reduction, exact Horner", horner(f2.astype(np.float64), False) / ref - 1)
show("everything in f32 (the split as shipped)", horner(f2.astype(np.float64), True) / ref - 1)
print("error of f itself (1 ulp of f = 3e-8 for |f| >= 1/4):")
show("first rounding   f1 - (u HI - n)", f1.astype(np.longdouble) - x1)
show("second rounding  f - (u LO + f1)", f2.astype(np.longdouble) - x2)
print("The second rounding adds u LO -- which moves by 0.06 ulp of f across one of these bins -- to a value that sits ON the f32 grid:"
      "\nit rounds the same way for every u nearby (up to half an ulp of f = 0.17 ulp of q), a sawtooth in u with a period of ~1.5.")
