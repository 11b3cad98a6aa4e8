#!/usr/bin/env python3
"""The reference's batch experiment (libepic/tests/batch/batch.py:52-164) on this library: per domain, the legacy CPU
SOR (double, omega = 1.5), the CPU log-space Gauss-Seidel and the GPU log-space solver, all at precision 1e-3 as there,
written as the same CSV (time per update, time to converge).  The reference's "Percent Valid" column comes from a
gradient flood fill in its compare_precision.py; here it is the share of sampled free cells whose streamline reaches a
goal -- legacy field walked by harmonic_legacy_compute_path_2d_cpu, log-space field walked on the device in one batch
(epic_hip_compute_paths_2d_gpu).  Domains: the three reference maps held under tests/golden/maps.
    python tools/batch.py out.csv [--no-cpu]"""
import ctypes as ct
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from epic_amd import epic_harmonic as eh  # noqa: E402
from epic_amd.harmonic_map import HarmonicMap  # noqa: E402

E = eh._epic
PRECISION = 1e-3
SAMPLES = 2000
DOMAINS = [("Basic", "basic"), ("Maze", "maze"), ("UMass", "umass")]
PF, PD = ct.POINTER(ct.c_float), ct.POINTER(ct.c_double)


def sample_starts(image, rng):
    free = np.argwhere((image != 0) & (image != 255))
    pick = free[rng.choice(len(free), min(SAMPLES, len(free)), replace=False)]
    return pick[:, ::-1].astype(np.float32)  # (x, y)


def reaches_goal(img, x, y):
    """The streamline's last point lies in a goal pixel (NaN = the gradient vanished: the legacy field's known failure)."""
    if not (np.isfinite(x) and np.isfinite(y)):
        return 0
    ex, ey = int(x + 0.5), int(y + 0.5)
    return int(0 <= ex < img.shape[1] and 0 <= ey < img.shape[0] and img[ey, ex] == 255)


def cpu_sor(h, starts):
    """Legacy SOR on the linear-space field (batch.py:52-110): u = 1 everywhere, 0 at goals; obstacles and goals locked."""
    img = h.image
    rows, cols = img.shape
    locked = np.ascontiguousarray(((img == 0) | (img == 255)).astype(np.uint32).ravel())
    u = np.ascontiguousarray((1.0 - (img == 255)).astype(np.float64).ravel())
    it = ct.c_uint(0)
    t0 = time.time()
    rc = E.harmonic_legacy_sor_2d_double_cpu(cols, rows, PRECISION, 1.5, locked.ctypes.data_as(eh._UP),
                                             u.ctypes.data_as(PD), ct.byref(it))
    dt = time.time() - t0
    assert rc == 0
    ok = 0
    for x, y in starts:
        k, raw = ct.c_uint(0), PD()
        rc = E.harmonic_legacy_compute_path_2d_cpu(cols, rows, locked.ctypes.data_as(eh._UP), u.ctypes.data_as(PD),
                                                   float(x), float(y), 0.2, 0.4, 200000, 0, ct.byref(k), ct.byref(raw))
        if rc == 0:
            ok += reaches_goal(img, raw[2 * k.value - 2], raw[2 * k.value - 1])
            E.harmonic_legacy_free_path_cpu(ct.byref(raw))
    return ok / len(starts), dt / max(1, it.value), dt


def gpu_valid(h, starts):
    """Streamlines of all samples in one launch on the resident field of a solved map."""
    for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu,
               E.harmonic_initialize_locked_gpu):
        assert fn(h) == 0
    n, max_len = len(starts), 20000
    k = np.zeros(n, dtype=np.uint32)
    rc = np.zeros(n, dtype=np.int32)
    out = np.empty((n, 2 * max_len), dtype=np.float32)
    s = np.ascontiguousarray(starts)
    assert E.epic_hip_compute_paths_2d_gpu(h, n, s.ctypes.data_as(PF), 0.2, 0.4, max_len, k.ctypes.data_as(eh._UP),
                                           rc.ctypes.data_as(ct.POINTER(ct.c_int)), out.ctypes.data_as(PF)) == 0
    for fn in (E.harmonic_uninitialize_dimension_size_gpu, E.harmonic_uninitialize_potential_values_gpu,
               E.harmonic_uninitialize_locked_gpu):
        fn(h)
    return sum(reaches_goal(h.image, out[i, 2 * k[i] - 2], out[i, 2 * k[i] - 1]) for i in range(n) if rc[i] == 0) / n


def main():
    if len(sys.argv) < 2:
        sys.exit("Please specify an output filename.")
    with_cpu = "--no-cpu" not in sys.argv
    rng = np.random.default_rng(2016)
    with open(sys.argv[1], "w") as f:
        f.write(",,,CPU SOR,,CPU log-GS,,GPU log-GS,,\n")
        f.write("Domain,Size,Percent Valid,Time per Update,Time to Converge,Time per Update,Time to Converge,"
                "Time per Update,Time to Converge,Percent Valid\n")
        for name, stem in DOMAINS:
            path = os.path.join(ROOT, "tests", "golden", "maps", stem + ".png")
            h = HarmonicMap().load(path)
            starts = sample_starts(h.image, rng)
            f.write("%s,%i," % (name, h.image.size))
            valid, per, total = cpu_sor(h, starts)
            f.write("%.5f,%.5f,%.5f," % (valid, per, total))
            f.flush()
            if with_cpu:
                h = HarmonicMap().load(path)
                wall, _ = h.solve(process="cpu", epsilon=PRECISION)
                f.write("%.5f,%.5f," % (wall / int(h.currentIteration), wall))
            else:
                f.write(",,")
            f.flush()
            HarmonicMap().load(path).solve(process="gpu", epsilon=PRECISION)   # code load, graph capture
            h = HarmonicMap().load(path)
            wall, _ = h.solve(process="gpu", epsilon=PRECISION)
            f.write("%.7f,%.5f," % (wall / int(h.currentIteration), wall))
            f.write("%.5f\n" % gpu_valid(h, starts))
            f.flush()
            print(".", end="", flush=True)
    print("\nDone.")


if __name__ == "__main__":
    main()
