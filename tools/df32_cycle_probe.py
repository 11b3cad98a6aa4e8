"""GPU probe: does the df32 Jacobi iteration settle on a fixed point or a short cycle on the reference maps?"""
import ctypes as ct, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O
from epic_amd import epic_harmonic as eh
from epic_amd.harmonic import Harmonic
from epic_amd.harmonic_map import HarmonicMap
E = eh._epic
g = np.load(os.path.join(ROOT, "tests/golden/maps_converged.npz"))
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for name in ("basic", "maze", "umass"):
    hm = HarmonicMap().load(os.path.join(ROOT, "tests/golden/maps", name + ".png"))
    h = Harmonic(); h.set_grid(list(hm.shape), hm.u_array(), hm.locked_array()); h.epsilon = 1e-6
    for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
        assert fn(h) == 0
    assert E.harmonic_initialize_gpu(h, 1024) == 0
    E.epic_hip_set_math_mode(h, mode)
    want = g[name + "/converged_1e-06"]; free = hm.locked_array().ravel() == 0
    prev = None
    for chunk in range(200):
        rc = E.epic_hip_update_n_gpu(h, 999, 0); assert rc == 0
        E.harmonic_get_potential_values_gpu(h); a = h.u_array().ravel().copy()
        rc = E.epic_hip_update_n_gpu(h, 1, 1); d1 = h.delta
        E.harmonic_get_potential_values_gpu(h); b = h.u_array().ravel().copy()
        E.epic_hip_update_n_gpu(h, 1, 1); d2 = h.delta
        E.harmonic_get_potential_values_gpu(h); c = h.u_array().ravel().copy()
        h.currentIteration -= 2  # keep 1000-sweep cadence readable
        ncyc = int((a != c).sum()); nflip = int((a != b).sum())
        err = np.abs(c[free] - want[free]) / np.maximum(1, np.abs(want[free]))
        if chunk % 10 == 9 or (d1 < 1e-6) or ncyc == 0:
            print(f"{name} sweeps {(chunk+1)*1000:6d} delta {d1:.3e}/{d2:.3e} cells changing per sweep {nflip} period-2-different {ncyc} max rel err {err.max():.2e}", flush=True)
        if d1 < 1e-6 or ncyc == 0:
            break
