"""``Harmonic``: the solver object of the reference's python package (libepic/python/epic/harmonic.py:36-125),
bound to the MI355X-native library.

Same constructor defaults (epsilon 1e-2, stagger 100) and the same ``solve(algorithm, process, numThreads, epsilon)``
call sequence -- initialize x3 -> harmonic_complete_gpu -> uninitialize x3 (harmonic.py:67-92).  One deliberate
difference: the reference silently re-runs on the CPU when the GPU path fails (harmonic.py:76-86); here that is an
error unless the caller opts in with ``allow_cpu_fallback=True``, so a missing GPU can never masquerade as a GPU
result.
"""
import ctypes as ct
import time

import numpy as np

from . import epic_harmonic as eh


class Harmonic(eh.EpicHarmonic):
    """A log-space harmonic function over an n-D occupancy grid."""

    def __init__(self):
        self.n = 0
        self.m = ct.POINTER(ct.c_uint)()
        self.u = ct.POINTER(ct.c_float)()
        self.locked = ct.POINTER(ct.c_uint)()
        self.epsilon = 1e-2
        self.delta = self.epsilon + 1.0
        self.numIterationsToStaggerCheck = 100
        self.currentIteration = 0
        self.d_m = ct.POINTER(ct.c_uint)()
        self.d_u = ct.POINTER(ct.c_float)()
        self.d_locked = ct.POINTER(ct.c_uint)()
        self.d_delta = ct.POINTER(ct.c_float)()
        self._keep = {}

    # -- data ----------------------------------------------------------------------------------------------
    def set_grid(self, m, u, locked):
        """Attach host arrays (copied into numpy arrays this object keeps alive)."""
        m = np.ascontiguousarray(np.asarray(m, dtype=np.uint32))
        u = np.ascontiguousarray(np.asarray(u, dtype=np.float32)).reshape(-1).copy()
        locked = np.ascontiguousarray(np.asarray(locked, dtype=np.uint32)).reshape(-1).copy()
        cells = int(np.prod(m.astype(np.int64)))
        if u.size != cells or locked.size != cells:
            raise ValueError("u and locked must have prod(m) = %d elements" % cells)
        self._keep = dict(m=m, u=u, locked=locked)
        self.n = len(m)
        self.m = m.ctypes.data_as(ct.POINTER(ct.c_uint))
        self.u = u.ctypes.data_as(ct.POINTER(ct.c_float))
        self.locked = locked.ctypes.data_as(ct.POINTER(ct.c_uint))

    @property
    def shape(self):
        return tuple(int(self.m[i]) for i in range(self.n))

    def u_array(self):
        """The host potential values as a numpy view of shape m."""
        if "u" in self._keep:
            return self._keep["u"].reshape(self.shape)
        cells = int(np.prod(self.shape))
        return np.ctypeslib.as_array(self.u, shape=(cells,)).reshape(self.shape)

    def locked_array(self):
        if "locked" in self._keep:
            return self._keep["locked"].reshape(self.shape)
        cells = int(np.prod(self.shape))
        return np.ctypeslib.as_array(self.locked, shape=(cells,)).reshape(self.shape)

    # -- solve ---------------------------------------------------------------------------------------------
    def solve(self, algorithm='gauss-seidel', process='gpu', numThreads=1024, epsilon=1e-2,
              allow_cpu_fallback=False):
        """Relax to ``epsilon``.  Returns (wall-time, cpu-time) of the solver call, excluding (un)initialisation,
        like the reference (harmonic.py:54-107).  ``process`` is 'gpu' (Jacobi sweeps on the MI355X) or 'cpu'
        (the exported red-black Gauss-Seidel)."""
        if algorithm != 'gauss-seidel':
            raise ValueError("the algorithm '%s' is undefined" % algorithm)
        self.epsilon = epsilon
        timing = None
        if process == 'gpu':
            result = eh._epic.harmonic_initialize_dimension_size_gpu(self)
            result += eh._epic.harmonic_initialize_potential_values_gpu(self)
            result += eh._epic.harmonic_initialize_locked_gpu(self)
            failed = result != 0
            if not failed:
                timing = (time.time(), time.process_time())
                result = eh._epic.harmonic_complete_gpu(self, int(numThreads))
                timing = (time.time() - timing[0], time.process_time() - timing[1])
                failed = result != 0
            # complete_gpu uninitialises on success; on failure (and for the re-entrant first set) do it here
            eh._epic.harmonic_uninitialize_dimension_size_gpu(self)
            eh._epic.harmonic_uninitialize_potential_values_gpu(self)
            eh._epic.harmonic_uninitialize_locked_gpu(self)
            eh._epic.harmonic_uninitialize_gpu(self)
            if failed:
                if not allow_cpu_fallback:
                    raise RuntimeError("epic_amd: the GPU solver failed with code %d and allow_cpu_fallback is off"
                                       % result)
                print("Failed to execute the 'epic' library's GPU solver; falling back to the CPU as requested.")
                process = 'cpu'
        if process == 'cpu':
            timing = (time.time(), time.process_time())
            result = eh._epic.harmonic_complete_cpu(self)
            timing = (time.time() - timing[0], time.process_time() - timing[1])
            if result != 0:
                raise RuntimeError("epic_amd: the CPU solver failed with code %d" % result)
        elif timing is None:
            raise ValueError("process must be 'gpu' or 'cpu'")
        return timing

    def __str__(self):
        s = "n: %d\nm: %s\nepsilon: %g\ndelta: %g\nnumIterationsToStaggerCheck: %d\ncurrentIteration: %d\n" % (
            self.n, list(self.shape), self.epsilon, self.delta, self.numIterationsToStaggerCheck,
            self.currentIteration)
        return s
