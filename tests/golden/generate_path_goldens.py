#!/usr/bin/env python3
"""Golden vectors for the NEXT rows: streamlines / potential / gradient on converged fields and
the legacy linear-space SOR, produced by RUNNING THE REFERENCE (oracle/_ref/libepic_ref.so = the reference's own sources,
compiled by oracle/Makefile).  Only the outputs (tests/golden/paths.npz) are read by the tests.

Inputs: the converged reference fields already committed in tests/golden/maps_converged.npz (so this runs in seconds).
Usage: python tests/golden/generate_path_goldens.py
"""
import ctypes as ct
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402

import hashlib


def store_path(out, key, pts):
    """Paths run to 7e4 points; commit their length, a SHA-256 of the raw bytes and both ends instead of every point."""
    out[key + "_k"] = np.int32(pts.size // 2)
    out[key + "_sha256"] = np.frombuffer(hashlib.sha256(pts.tobytes()).digest(), dtype=np.uint8)
    out[key + "_head"] = pts[:16].copy()
    out[key + "_tail"] = pts[-16:].copy()


F, D, U = ct.c_float, ct.c_double, ct.c_uint
PF, PD, PU = ct.POINTER(F), ct.POINTER(D), ct.POINTER(U)


def main():
    ref = O.ref()
    if ref is None:
        sys.exit("needs the compiled reference (oracle/_ref)")
    H = ct.POINTER(O.CHarmonic)
    ref.harmonic_compute_potential_2d_cpu.argtypes = (H, F, F, PF)
    ref.harmonic_compute_gradient_2d_cpu.argtypes = (H, F, F, F, PF, PF)
    ref.harmonic_compute_path_2d_cpu.argtypes = (H, F, F, F, F, U, PU, ct.POINTER(PF))
    ref.harmonic_free_path_cpu.argtypes = (ct.POINTER(PF),)
    ref.harmonic_legacy_sor_2d_float_cpu.argtypes = (U, U, F, F, PU, PF, PU)
    ref.harmonic_legacy_sor_2d_double_cpu.argtypes = (U, U, D, D, PU, PD, PU)
    ref.harmonic_legacy_sor_2d_long_double_cpu.argtypes = (U, U, ct.c_longdouble, ct.c_longdouble, PU,
                                                           ct.POINTER(ct.c_longdouble), PU)
    ref.harmonic_legacy_compute_potential_2d_cpu.argtypes = (U, U, PU, PD, D, D, PD)
    ref.harmonic_legacy_compute_gradient_2d_cpu.argtypes = (U, U, PU, PD, D, D, D, PD, PD)
    ref.harmonic_legacy_compute_path_2d_cpu.argtypes = (U, U, PU, PD, D, D, D, D, U, ct.c_int, PU, ct.POINTER(PD))
    ref.harmonic_legacy_free_path_cpu.argtypes = (ct.POINTER(PD),)

    maps = np.load(os.path.join(HERE, "maps_converged.npz"))
    out = {}
    rng = np.random.default_rng(2024)
    for name in ("basic", "umass", "maze"):
        m, u0, locked = O.load_png_reference_rule(os.path.join(HERE, "maps", name + ".png"))
        p = O.Problem(m, maps[name + "/converged_1e-06"], locked)
        rows, cols = m
        free = np.argwhere((locked.reshape(m) == 0))
        # probes: potential + gradient at random sub-cell positions inside free cells (incl. some that fail)
        pick = free[rng.choice(len(free), size=64, replace=False)]
        xs = (pick[:, 1] + rng.uniform(-0.45, 0.45, 64)).astype(np.float32)
        ys = (pick[:, 0] + rng.uniform(-0.45, 0.45, 64)).astype(np.float32)
        xs[:4] = [-1.0, cols + 3.0, 0.2, cols - 1.0]          # outside / on the locked border
        pot = np.zeros(64, np.float32)
        prc = np.zeros(64, np.int32)
        gx = np.zeros(64, np.float32)
        gy = np.zeros(64, np.float32)
        grc = np.zeros(64, np.int32)
        for i in range(64):
            v, a, b = F(0), F(0), F(0)
            prc[i] = ref.harmonic_compute_potential_2d_cpu(ct.byref(p.h), xs[i], ys[i], ct.byref(v))
            pot[i] = v.value
            grc[i] = ref.harmonic_compute_gradient_2d_cpu(ct.byref(p.h), xs[i], ys[i], 0.5, ct.byref(a), ct.byref(b))
            gx[i], gy[i] = a.value, b.value
        out.update({f"{name}/probe_x": xs, f"{name}/probe_y": ys, f"{name}/pot": pot, f"{name}/pot_rc": prc,
                    f"{name}/gx": gx, f"{name}/gy": gy, f"{name}/grad_rc": grc})
        # streamlines: python wrapper parameters (harmonic_map.py:117: step 0.2, cd 0.4, maxLength 1e6) and the
        # plugin's (epic_nav_core_plugin.cpp:291-298: step 0.05, cd 0.5)
        starts = free[rng.choice(len(free), size=6, replace=False)]
        for j, (sy, sx) in enumerate(starts):
            step, cd = (0.2, 0.4) if j % 2 == 0 else (0.05, 0.5)
            k, raw = U(0), PF()
            rc = ref.harmonic_compute_path_2d_cpu(ct.byref(p.h), float(sx), float(sy), step, cd, 1000000, ct.byref(k),
                                                  ct.byref(raw))
            pts = np.ctypeslib.as_array(raw, shape=(2 * k.value,)).copy() if rc == 0 else np.zeros(0, np.float32)
            if rc == 0:
                ref.harmonic_free_path_cpu(ct.byref(raw))
            out[f"{name}/path{j}_start"] = np.array([sx, sy, step, cd], np.float32)
            out[f"{name}/path{j}_rc"] = np.int32(rc)
            store_path(out, f"{name}/path{j}", pts)
            print(f"  {name} path {j}: start ({sx},{sy}) step {step} rc {rc} k {k.value}")

    # legacy linear-space SOR on a small seeded grid: u = 1 obstacles/border, 0 goal (flipped = 0 convention)
    w, h = 40, 28
    _, locked = O.oracle_synthetic([h, w], 31, 0.08)
    u0 = np.ones(h * w)
    goal = (h // 2) * w + w // 2
    u0[goal] = 0.0
    for tag, ctype, dtype, fn in (("float", F, np.float32, ref.harmonic_legacy_sor_2d_float_cpu),
                                  ("double", D, np.float64, ref.harmonic_legacy_sor_2d_double_cpu),
                                  ("long_double", ct.c_longdouble, np.longdouble, ref.harmonic_legacy_sor_2d_long_double_cpu)):
        u = u0.astype(dtype)
        it = U(0)
        lk = locked.copy()
        rc = fn(w, h, ctype(1e-3), ctype(1.5), lk.ctypes.data_as(PU), u.ctypes.data_as(ct.POINTER(ctype)), ct.byref(it))
        assert rc == 0
        out[f"legacy/{tag}_u"] = u.astype(np.float64)   # long double narrowed for storage; compared with a tolerance
        out[f"legacy/{tag}_iter"] = np.int32(it.value)
        print(f"  legacy {tag}: {it.value} iterations")
    out["legacy/w_h"] = np.array([w, h], np.int32)
    out["legacy/locked"] = locked.astype(np.uint8)
    out["legacy/u0"] = u0
    # legacy path on the double field
    ud = out["legacy/double_u"].copy()
    lk = locked.copy()
    for j, (sx, sy) in enumerate(((5.0, 5.0), (33.0, 20.0))):
        k, raw = U(0), PD()
        rc = ref.harmonic_legacy_compute_path_2d_cpu(w, h, lk.ctypes.data_as(PU), ud.ctypes.data_as(PD), sx, sy, 0.2, 0.4,
                                                     4000, 0, ct.byref(k), ct.byref(raw))
        pts = np.ctypeslib.as_array(raw, shape=(2 * k.value,)).copy() if rc == 0 else np.zeros(0)
        if rc == 0:
            ref.harmonic_legacy_free_path_cpu(ct.byref(raw))
        out[f"legacy/path{j}_start"] = np.array([sx, sy])
        out[f"legacy/path{j}_rc"] = np.int32(rc)
        store_path(out, f"legacy/path{j}", pts)
        print(f"  legacy path {j}: rc {rc} k {k.value}")
    np.savez_compressed(os.path.join(HERE, "paths.npz"), **out)
    print("wrote paths.npz")


if __name__ == "__main__":
    main()
