"""How many cells still change per Jacobi sweep as the 8192^2 relaxation proceeds?  (Feasibility of skipping tiles whose
inputs did not change: their update is bit-for-bit a no-op.)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from epic_amd import epic_harmonic as eh
from epic_amd.harmonic import Harmonic
from epic_amd.synthetic import synthetic_grid
E = eh._epic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
u0, locked = synthetic_grid([n, n])
h = Harmonic(); h.set_grid([n, n], u0, locked); h.epsilon = 1e-6
for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
    assert fn(h) == 0
assert E.harmonic_initialize_gpu(h, 1024) == 0
done = 0
for target in (1000, 4000, 8000, 12000, 16000, 20000, 25000, 30000, 35000, 40000, 44000):
    E.epic_hip_update_n_gpu(h, target - done - 2, 0); done = target - 2
    E.harmonic_get_potential_values_gpu(h); a = h.u_array().copy()
    E.epic_hip_update_n_gpu(h, 2, 1); done += 2
    E.harmonic_get_potential_values_gpu(h); b = h.u_array()
    ch = a != b
    # 64 x 256 tiles (a wave-task): fraction with any changed cell
    t = ch.reshape(n // 64, 64, n // 256, 256).any(axis=(1, 3))
    print(f"sweep {target:6d}: delta {h.delta:.3e}  cells changed over 2 sweeps {ch.mean()*100:6.2f} %  64x256 tiles touched {t.mean()*100:6.2f} %", flush=True)
