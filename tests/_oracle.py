"""ctypes access to the CHECKER libraries (test infrastructure only).

* ``oracle/liboracle.so``        -- the plain-C restatement (oracle/harmonic_oracle.c); travels to the GPU box.
* ``oracle/_ref/libepic_ref.so`` -- the reference's own CPU sources compiled by oracle/Makefile; exists only
  where /root/reference was available at build time (it is prebuilt into the snapshot for the GPU box).

Nothing under epic_amd/ imports this module.
"""
import ctypes as ct
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libepic_ref.so")


class CHarmonic(ct.Structure):
    """Field order/types of the reference struct (libepic/include/epic/harmonic/harmonic.h:44-64,
    ctypes twin libepic/python/epic/epic_harmonic.py:45-57)."""

    _fields_ = [
        ("n", ct.c_uint),
        ("m", ct.POINTER(ct.c_uint)),
        ("u", ct.POINTER(ct.c_float)),
        ("locked", ct.POINTER(ct.c_uint)),
        ("epsilon", ct.c_float),
        ("delta", ct.c_float),
        ("numIterationsToStaggerCheck", ct.c_uint),
        ("currentIteration", ct.c_uint),
        ("d_m", ct.POINTER(ct.c_uint)),
        ("d_u", ct.POINTER(ct.c_float)),
        ("d_locked", ct.POINTER(ct.c_uint)),
        ("d_delta", ct.POINTER(ct.c_float)),
    ]


class Problem:
    """Owns the numpy arrays a CHarmonic points into."""

    def __init__(self, m, u, locked, epsilon=1e-6, stagger=100):
        self.m = np.ascontiguousarray(np.asarray(m, dtype=np.uint32))
        self.u = np.ascontiguousarray(np.asarray(u, dtype=np.float32)).reshape(-1).copy()
        self.locked = np.ascontiguousarray(np.asarray(locked, dtype=np.uint32)).reshape(-1).copy()
        assert self.u.size == int(np.prod(self.m.astype(np.int64))) == self.locked.size
        self.h = CHarmonic()
        self.h.n = len(self.m)
        self.h.m = self.m.ctypes.data_as(ct.POINTER(ct.c_uint))
        self.h.u = self.u.ctypes.data_as(ct.POINTER(ct.c_float))
        self.h.locked = self.locked.ctypes.data_as(ct.POINTER(ct.c_uint))
        self.h.epsilon = epsilon
        self.h.delta = epsilon + 1.0
        self.h.numIterationsToStaggerCheck = stagger
        self.h.currentIteration = 0

    @property
    def shape(self):
        return tuple(int(x) for x in self.m)

    def field(self):
        return self.u.reshape(self.shape)

    def clone(self, **kw):
        return Problem(self.m, self.u, self.locked, kw.get("epsilon", float(self.h.epsilon)),
                       kw.get("stagger", int(self.h.numIterationsToStaggerCheck)))


def build_oracle():
    """(Re)build the checker libraries; building the checker is not using it."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        lib = ct.CDLL(ORACLE_SO)
        P = ct.POINTER(CHarmonic)
        for name in ("oracle_update", "oracle_update_and_check", "oracle_complete", "oracle_jacobi_complete"):
            getattr(lib, name).argtypes = (P,)
            getattr(lib, name).restype = ct.c_int
        lib.oracle_jacobi_run.argtypes = (P, ct.c_uint)
        lib.oracle_set_cells_2d.argtypes = (P, ct.c_uint, ct.POINTER(ct.c_uint), ct.POINTER(ct.c_uint))
        lib.oracle_synthetic.argtypes = (ct.c_uint, ct.POINTER(ct.c_uint), ct.c_uint64, ct.c_double,
                                         ct.POINTER(ct.c_float), ct.POINTER(ct.c_uint))
        lib.oracle_synthetic.restype = None
        lib.oracle_scramble_free.argtypes = (ct.c_uint, ct.POINTER(ct.c_uint), ct.c_uint64, ct.c_float, ct.c_float,
                                             ct.POINTER(ct.c_float), ct.POINTER(ct.c_uint))
        lib.oracle_scramble_free.restype = None
        lib.oracle_cell_updates.restype = ct.c_uint64
        lib.oracle_reset_counters.restype = None
        lib.oracle_tol_run.argtypes = (P, ct.c_uint, ct.c_int)      # oracle/tol_checker.c
        lib.oracle_tol_complete.argtypes = (P, ct.c_int)
        lib.oracle_tol_set_finish.argtypes = (ct.c_int,)
        lib.oracle_tol_set_finish.restype = None
        lib.oracle_tol_split.argtypes = (ct.POINTER(ct.c_float), ct.c_size_t, ct.POINTER(ct.c_float), ct.POINTER(ct.c_int))
        lib.oracle_tol_split.restype = None
        lib.oracle_libm_mismatches.restype = ct.c_size_t
        lib.oracle_libm_mismatches.argtypes = (ct.c_int, ct.c_uint32, ct.c_size_t, ct.c_void_p, ct.POINTER(ct.c_size_t),
                                               ct.c_int)
        _oracle = lib
    return _oracle


def ref():
    """The compiled reference, or None when it was never built (no /root/reference)."""
    global _ref
    if _ref is None:
        if not os.path.exists(REF_SO):
            if os.path.isdir("/root/reference"):
                build_oracle()
            if not os.path.exists(REF_SO):
                return None
        lib = ct.CDLL(REF_SO)
        P = ct.POINTER(CHarmonic)
        for name in ("harmonic_complete_cpu", "harmonic_update_cpu", "harmonic_update_and_check_cpu"):
            getattr(lib, name).argtypes = (P,)
            getattr(lib, name).restype = ct.c_int
        lib.harmonic_utilities_set_cells_2d_cpu.argtypes = (P, ct.c_uint, ct.POINTER(ct.c_uint),
                                                            ct.POINTER(ct.c_uint))
        _ref = lib
    return _ref


def oracle_synthetic(m, seed=20240601, density=0.05):
    m = np.asarray(m, dtype=np.uint32)
    cells = int(np.prod(m.astype(np.int64)))
    u = np.empty(cells, dtype=np.float32)
    locked = np.empty(cells, dtype=np.uint32)
    oracle().oracle_synthetic(len(m), m.ctypes.data_as(ct.POINTER(ct.c_uint)), seed, density,
                              u.ctypes.data_as(ct.POINTER(ct.c_float)),
                              locked.ctypes.data_as(ct.POINTER(ct.c_uint)))
    return u, locked


def scramble_free(m, u, locked, seed=77, lo=-50.0, hi=0.0):
    """In place: every unlocked cell of u (float32, flat) gets a seeded value in [lo, hi) -- see oracle_scramble_free."""
    m = np.asarray(m, dtype=np.uint32)
    assert u.dtype == np.float32 and locked.dtype == np.uint32 and u.flags.c_contiguous and locked.flags.c_contiguous
    oracle().oracle_scramble_free(len(m), m.ctypes.data_as(ct.POINTER(ct.c_uint)), seed, lo, hi,
                                  u.ctypes.data_as(ct.POINTER(ct.c_float)), locked.ctypes.data_as(ct.POINTER(ct.c_uint)))
    return u


def load_png_reference_rule(path):
    """The reference's python loader rule (libepic/python/epic/harmonic_map.py:70-100) restated with PIL:
    m = [H, W]; u = 0.0 where pixel == 255 else -1e6; locked = pixel in {0, 255}."""
    from PIL import Image

    px = np.array(Image.open(path).convert("L"))
    u = np.where(px == 255, np.float32(0.0), np.float32(-1e6)).astype(np.float32)
    locked = ((px == 0) | (px == 255)).astype(np.uint32)
    return [px.shape[0], px.shape[1]], u, locked


def session_scheme():
    """The iteration scheme a library context created NOW runs: EPIC_HIP_SCHEME as the library reads it (absent: its default,
    the reference's red-black half-sweeps).  tests/conftest.py sets the variable for the session (EPIC_TEST_SCHEME)."""
    return "jacobi" if os.environ.get("EPIC_HIP_SCHEME") == "jacobi" else "redblack"


def run_session(p, k):
    """k iterations of the session's scheme on Problem p with the reference's arithmetic, from p.h.currentIteration on; the last
    one is a check (p.h.delta = its max |du|) -- what k - 1 harmonic_update_gpu + one harmonic_update_and_check_gpu of a
    context without an explicit epic_hip_set_scheme must reproduce."""
    lib = oracle()
    if session_scheme() == "jacobi":
        return lib.oracle_jacobi_run(ct.byref(p.h), k)
    for i in range(k):
        (lib.oracle_update_and_check if i == k - 1 else lib.oracle_update)(ct.byref(p.h))
    return 0
