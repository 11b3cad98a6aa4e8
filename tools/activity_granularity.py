"""Library defaults (precise red-black), 8192^2: how much of what the work lists recompute actually changes?  Cells that change
over two iterations, and the share of tiles of several shapes that hold at least one such cell (the lists work on 16 x 256;
finer tiles would recompute less, if the changing cells are clustered)."""
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
SHAPES = [(16, 256), (8, 256), (4, 256), (16, 64), (8, 64), (4, 64), (16, 16)]
print("iteration   delta     cells  " + "  ".join("%dx%d" % s for s in SHAPES))
for target in (2000, 6000, 10000, 14000, 18000, 22000, 26000, 30000, 34000, 38000, 42000, 44800):
    E.epic_hip_update_n_gpu(h, target - done - 2, 0); done = target - 2
    E.harmonic_get_potential_values_gpu(h); a = torch.from_numpy(h.u_array().copy()).cuda()
    E.epic_hip_update_n_gpu(h, 2, 1); done += 2
    E.harmonic_get_potential_values_gpu(h); b = torch.from_numpy(h.u_array()).cuda()
    ch = (a.view(torch.int32) != b.view(torch.int32))
    row = [ch.float().mean().item()]
    for r, c in SHAPES:
        row.append(ch.view(n // r, r, n // c, c).any(dim=3).any(dim=1).float().mean().item())
    print(f"{target:8d}  {h.delta:.2e}  " + "  ".join("%6.2f" % (100 * x) for x in row), flush=True)
