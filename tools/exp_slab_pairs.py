"""8192^2 default relaxation (precise, red-black, work lists) on one device and on in-library slabs of the same GPU, with the work lists
always bypassed (EPIC_HIP_TRACK_SWITCH=0), never (2) and by the rule: where do the slabs lose time?  (round 6)"""
import os

os.environ.setdefault("EPIC_HIP_STUDY", "1")   # this tool steers the kernel plan with study knobs (epic_amd/csrc/driver_config.cpp)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epic_amd import epic_harmonic as eh
from epic_amd.harmonic import Harmonic
from epic_amd.synthetic import synthetic_grid

E = eh._epic
m = [8192, 8192]
u0, locked = synthetic_grid(m)
for switch in (None, "0", "2"):
    for devs in (None, "0,0", "0,0,0,0"):
        for extra in ((None,), ("EPIC_HIP_THREADS",)) if devs else ((None,),):
            env = {"EPIC_HIP_TRACK_SWITCH": switch, "EPIC_HIP_DEVICES": devs}
            if extra[0]:
                env[extra[0]] = "0"
            for k, v in env.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
            h = Harmonic()
            h.set_grid(m, u0, locked)
            h.epsilon, h.numIterationsToStaggerCheck = 1e-6, 100
            for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
                assert fn(h) == 0
            t0 = time.perf_counter()
            assert E.harmonic_execute_gpu(h, 1024) == 0
            dt = time.perf_counter() - t0
            for fn in (E.harmonic_uninitialize_dimension_size_gpu, E.harmonic_uninitialize_potential_values_gpu, E.harmonic_uninitialize_locked_gpu):
                fn(h)
            for k in list(env) + ["EPIC_HIP_THREADS"]:
                os.environ.pop(k, None)
            print("switch %-4s devices %-8s %-12s %d iterations %.3f s" % (switch, devs, "caller issues" if extra[0] else "", h.currentIteration, dt), flush=True)
