"""Per-iteration time of the tol arithmetic on slab-sized grids (1024 .. 8192 rows x 8192 columns: what one GPU of 8, 4, 2, 1
holds in the strong-scaling run), through epic_amd/slab.py (world = 1: no exchange) and through the C-ABI -- host wall clock and
device time.  Shows what the Python driver loop costs (nothing: the GPU is the bottleneck from 1024 rows up) and how far a
short slab is from 1/N of the full grid.  Run on the GPU box: python tools/time_slab_sizes.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from epic_amd.slab import SlabSolver
import ctypes as ct
from epic_amd import epic_harmonic as eh
from epic_amd.harmonic import Harmonic
from epic_amd.synthetic import synthetic_grid
E = eh._epic
for rows in (1024, 2048, 4096, 8192):
    grid = [rows, 8192]
    s = SlabSolver(grid, 0, 1, device=torch.device("cuda:0"), stagger=100, math="tol")
    s.load_synthetic()
    for _ in range(3): s.timed_step()
    torch.cuda.synchronize(); t0 = time.perf_counter(); dev = 0.0
    for _ in range(10): dev += s.timed_step()
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    del s
    u0, locked = synthetic_grid(grid)
    h = Harmonic(); h.set_grid(grid, u0, locked); h.epsilon = 1e-6; h.numIterationsToStaggerCheck = 100
    for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
        assert fn(h) == 0
    assert E.harmonic_initialize_gpu(h, 1024) == 0
    assert E.epic_hip_set_math_mode(h, 4) == 0 and E.epic_hip_set_activity_tracking(h, 0) == 0
    ms = ct.c_float(0)
    for _ in range(3): E.epic_hip_timed_sweeps_gpu(h, 100, 100, ct.byref(ms))
    torch.cuda.synchronize(); t0 = time.perf_counter(); adev = 0.0
    for _ in range(10):
        E.epic_hip_timed_sweeps_gpu(h, 100, 100, ct.byref(ms)); adev += ms.value
    torch.cuda.synchronize(); awall = time.perf_counter() - t0
    print(f"{grid}: slab.py wall {wall*1e3:.1f} us/iter, device {dev:.1f} us/iter | C-ABI wall {awall*1e3:.1f} us/iter, device {adev:.1f} us/iter", flush=True)
