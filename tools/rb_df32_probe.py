"""Does red-black + df32 satisfy the reference's convergence test, and how close to the reference field does it land?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["EPIC_HIP_SCHEME"] = "redblack"; os.environ["EPIC_HIP_MATH"] = "df32"
from epic_amd import epic_harmonic as eh
from epic_amd.harmonic import Harmonic
from epic_amd.harmonic_map import HarmonicMap
E = eh._epic
g = np.load(os.path.join(ROOT, "tests/golden/maps_converged.npz"))
for name in ("basic", "maze", "umass"):
    hm = HarmonicMap().load(os.path.join(ROOT, "tests/golden/maps", name + ".png"))
    h = Harmonic(); h.set_grid(list(hm.shape), hm.u_array(), hm.locked_array()); h.epsilon = 1e-6
    for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
        assert fn(h) == 0
    assert E.harmonic_initialize_gpu(h, 1024) == 0
    want = g[name + "/converged_1e-06"]; free = hm.locked_array().ravel() == 0
    done = None
    for chunk in range(400):
        rc = E.epic_hip_update_n_gpu(h, 999, 0); rc = E.epic_hip_update_n_gpu(h, 1, 1)
        if rc == 1:
            done = h.currentIteration; break
    E.harmonic_get_potential_values_gpu(h); a = h.u_array().ravel()
    err = np.abs(a[free] - want[free]) / np.maximum(1, np.abs(want[free]))
    print(f"rb+df32 {name}: converged at {done}, delta {h.delta:.3e}, max rel err {err.max():.2e}", flush=True)
