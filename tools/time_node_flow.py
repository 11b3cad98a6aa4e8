"""The navigation node's call loop (tests/plugin_replay/node_flow.cpp: one harmonic_update_and_check_gpu + steps - 1 harmonic_update_gpu
per tick) on one of the reference's maps, library defaults, for a fixed number of iterations -- the command profiled by
tools/profile_round.sh (kernel trace: which kernels the fine-grained API reaches, how many launches per iteration).

    python tools/time_node_flow.py [--map maze] [--iterations 20000] [--steps 50]        (EPIC_HIP_DEFER=0: one launch per call)"""
import argparse
import ctypes as ct
import os

os.environ.setdefault("EPIC_HIP_STUDY", "1")   # this tool steers the kernel plan with study knobs (epic_amd/csrc/driver_config.cpp)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epic_amd import epic_harmonic as eh  # noqa: E402
from epic_amd.harmonic_map import HarmonicMap  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--map", default="maze")
ap.add_argument("--iterations", type=int, default=20000)
ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()
E = eh._epic
nf = ct.CDLL(os.path.join(ROOT, "tests", "plugin_replay", "libnodeflow.so"))
nf.node_flow_run.restype = ct.c_int
h = HarmonicMap().load(os.path.join(ROOT, "tests", "golden", "maps", a.map + ".png"))
h.epsilon, h.numIterationsToStaggerCheck = 1e-6, 100
for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
    assert fn(h) == 0
assert E.harmonic_initialize_gpu(h, 1024) == 0
sec, done, conv = ct.c_double(0.0), ct.c_uint(0), ct.c_uint(0)
assert nf.node_flow_run(ct.byref(h), a.iterations, a.steps, 1024, 1, ct.byref(sec), ct.byref(done), ct.byref(conv)) == 0
print("%s: %d iterations in ticks of %d: %.4f s, %.3f us per iteration (EPIC_HIP_DEFER=%s)" % (a.map, done.value, a.steps, sec.value, sec.value / done.value * 1e6,
                                                                                          os.environ.get("EPIC_HIP_DEFER", "unset")))
for fn in (E.harmonic_uninitialize_gpu, E.harmonic_uninitialize_dimension_size_gpu, E.harmonic_uninitialize_potential_values_gpu,
           E.harmonic_uninitialize_locked_gpu):
    fn(h)
