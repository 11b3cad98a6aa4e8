"""The linear-space SOR baseline of the reference's python package, kept so that code written against it still runs.

Mirrors ``libepic/python/epic/harmonic_legacy.py:34-95`` (``HarmonicLegacy``: w, h, epsilon, omega, locked, u,
currentIteration; ``solve(omega, epsilon)`` -> (wall, cpu) seconds through harmonic_legacy_sor_2d_double_cpu) and
``harmonic_legacy_map.py:38-123`` (``HarmonicLegacyMap``: load a grayscale map -- 0 obstacle, 255 goal with u = 0,
everything else free with u = 1 -- optional flipped potential, streamline by harmonic_legacy_compute_path_2d_cpu).
CPU only, as in the reference: the legacy solver is the paper's comparison baseline and is never accelerated
(SURVEY.md §8f row 4).  The OpenCV viewer of the reference class is out of scope.
"""
import ctypes as ct
import time

import numpy as np

from . import epic_harmonic as eh

_PD = ct.POINTER(ct.c_double)


class HarmonicLegacy(object):
    """A 2-d potential in linear space relaxed by successive over-relaxation on the host."""

    def __init__(self):
        self.w = 0
        self.h = 0
        self.epsilon = 1e-2
        self.omega = 1.0
        self.locked = ct.POINTER(ct.c_uint)()
        self.u = _PD()
        self.currentIteration = 0
        self._keep = {}

    def set_grid(self, u, locked):
        """u (h x w, float64: 1 free / obstacle, 0 goal) and locked (h x w, 0 / 1); the arrays are copied and owned."""
        u = np.ascontiguousarray(np.asarray(u, dtype=np.float64))
        locked = np.ascontiguousarray(np.asarray(locked, dtype=np.uint32))
        if u.ndim != 2 or u.shape != locked.shape:
            raise ValueError("u and locked must be 2-d arrays of the same shape")
        self.h, self.w = (int(v) for v in u.shape)
        self._keep = dict(u=u.copy(), locked=locked.copy())
        self.u = self._keep["u"].ctypes.data_as(_PD)
        self.locked = self._keep["locked"].ctypes.data_as(ct.POINTER(ct.c_uint))

    def u_array(self):
        return self._keep["u"] if "u" in self._keep else np.ctypeslib.as_array(self.u, shape=(self.h, self.w))

    def locked_array(self):
        return self._keep["locked"] if "locked" in self._keep else np.ctypeslib.as_array(self.locked, shape=(self.h, self.w))

    def solve(self, omega=1.0, epsilon=1e-2):
        """Relax until max |du| < epsilon.  Returns (wall-time, cpu-time) of the solver call."""
        self.epsilon = epsilon
        self.omega = omega
        iterations = ct.c_uint(0)
        t0 = (time.time(), time.process_time())
        rc = eh._epic.harmonic_legacy_sor_2d_double_cpu(self.w, self.h, float(self.epsilon), float(self.omega), self.locked,
                                                        self.u, ct.byref(iterations))
        timing = (time.time() - t0[0], time.process_time() - t0[1])
        self.currentIteration = int(iterations.value)
        if rc != 0:
            raise RuntimeError("harmonic_legacy_sor_2d_double_cpu failed with code %d" % rc)
        return timing

    def __str__(self):
        return ("w:        %d\nh:        %d\nepsilon:  %s\nomega:    %s\nlocked:\n%s\n\nu:\n%s\n\n"
                % (self.w, self.h, self.epsilon, self.omega, self.locked_array(), self.u_array()))


class HarmonicLegacyMap(HarmonicLegacy):
    """A ``HarmonicLegacy`` loaded from a grayscale image (the reference's 0 / 255 / other rule)."""

    def __init__(self):
        super().__init__()
        self.image = None
        self.originalImage = None
        self.flipped = False

    def load(self, filename):
        from PIL import Image

        self.image = np.array(Image.open(filename).convert("L"))
        self.originalImage = self.image.copy()
        u = 1.0 - (self.image == 255).astype(np.float64)
        if self.flipped:
            u = 1.0 - u
        locked = ((self.image == 0) | (self.image == 255)).astype(np.uint32)
        self.set_grid(u, locked)
        return self

    def _flip_u_values(self):
        self.u_array()[...] = 1.0 - self.u_array()

    def compute_streamline(self, x, y, step_size=0.2, cd_precision=0.4, max_length=int(1e6)):
        """Way-points from the "double pixel" (x, y) towards a goal, as a list of (x, y) tuples
        (harmonic_legacy_map.py:95-123: same defaults, `flipped` decides ascent or descent)."""
        k = ct.c_uint(0)
        raw = _PD()
        rc = eh._epic.harmonic_legacy_compute_path_2d_cpu(self.w, self.h, self.locked, self.u, float(x), float(y),
                                                          float(step_size), float(cd_precision), int(max_length),
                                                          int(self.flipped), ct.byref(k), ct.byref(raw))
        if rc != 0:
            raise RuntimeError("harmonic_legacy_compute_path_2d_cpu failed with code %d" % rc)
        path = [(raw[2 * i], raw[2 * i + 1]) for i in range(int(k.value))]
        eh._epic.harmonic_legacy_free_path_cpu(ct.byref(raw))
        return path

    _compute_streamline = compute_streamline
