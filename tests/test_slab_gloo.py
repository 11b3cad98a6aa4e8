"""The multi-GPU path (epic_amd/slab.py) exercised on CPU tensors with the gloo backend, world_size 2 and 3.

What is under test is the decomposition itself -- row partition, ghost rows, the per-sweep halo exchange, the MAX
all-reduce of delta, the driver loop -- so the per-row arithmetic is injected from the checker
(oracle_jacobi_rows_2d); on the GPU the same class runs with the HIP backend.  The distributed result must equal the
single-domain checker bit for bit."""
import ctypes as ct
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _oracle as O
from epic_amd.slab import SlabSolver, partition_rows, synthetic_rows
from epic_amd.synthetic import synthetic_grid


class OracleBackend:
    """Test-only sweep backend: plain row-major 'mask' (the locked words themselves), rows swept by the checker."""

    def __init__(self, pairs=False):
        self.pairs = pairs   # offer sweep2 (two sweeps as one pass), as the HIP backend does for the tol math
        self.lib = O.oracle()
        self.lib.oracle_jacobi_rows_2d.restype = ct.c_float
        self.lib.oracle_jacobi_rows_2d.argtypes = (ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_uint, ct.c_uint,
                                                   ct.c_uint, ct.c_uint, ct.c_uint)
        self.lib.oracle_redblack_rows_2d.restype = ct.c_float
        self.lib.oracle_redblack_rows_2d.argtypes = (ct.c_void_p, ct.c_void_p, ct.c_uint, ct.c_uint, ct.c_uint,
                                                     ct.c_uint, ct.c_uint, ct.c_uint)
        self.cols = None

    def pitch_for(self, cols):
        self.cols = cols
        return (cols + 63) // 64 * 64

    def mask_words(self, rows, pitch):
        return rows * self.cols

    def pack_mask(self, locked_i32, rows, cols, pitch, ghost_top, ghost_bottom, maskw):
        lk = locked_i32.clone().reshape(rows, cols)
        if ghost_top:
            lk[0] = 1
        if ghost_bottom:
            lk[rows - 1] = 1
        maskw.copy_(lk.reshape(-1))

    def sweep(self, src, dst, maskw, rows, pitch, row_begin, row_end, delta_bits):
        if row_end <= row_begin:
            return
        # the checker treats local rows 0 / rows-1 as fixed; ghost rows are forced locked and true border rows are
        # locked by construction, so that is exactly the slab semantics
        d = self.lib.oracle_jacobi_rows_2d(src.data_ptr(), dst.data_ptr(), maskw.data_ptr(), rows, self.cols, pitch,
                                           row_begin, row_end)
        if delta_bits is not None:
            cur = delta_bits.view(torch.float32)
            cur[0] = max(float(cur[0]), d)

    def sweep2(self, src, dst, maskw, rows, pitch):
        tmp = torch.empty_like(src)
        self.lib.oracle_jacobi_rows_2d(src.data_ptr(), tmp.data_ptr(), maskw.data_ptr(), rows, self.cols, pitch, 0, rows)
        self.lib.oracle_jacobi_rows_2d(tmp.data_ptr(), dst.data_ptr(), maskw.data_ptr(), rows, self.cols, pitch, 0, rows)

    def sweep_rb(self, u, maskw, rows, pitch, row_begin, row_end, parity, delta_bits):
        if row_end <= row_begin:
            return
        d = self.lib.oracle_redblack_rows_2d(u.data_ptr(), maskw.data_ptr(), rows, self.cols, pitch, row_begin, row_end,
                                             parity)
        if delta_bits is not None:
            cur = delta_bits.view(torch.float32)
            cur[0] = max(float(cur[0]), d)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, grid, seed, sweeps, mode, out_dir, halo=8, scheme="jacobi", problem=None, pairs=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stagger, eps = (10, 1e-6) if problem is None else (problem["stagger"], problem["epsilon"])
        s = SlabSolver(grid, rank, world, device="cpu", stagger=stagger, epsilon=eps, backend=OracleBackend(pairs), halo=halo,
                       scheme=scheme)
        if problem is None:
            free = s.load_synthetic(seed=seed, density=0.08)
        else:
            top, bot = s.lo - s.g_top, s.hi + s.g_bot
            free = s.load_rows(problem["u0"].reshape(grid)[top:bot], problem["locked"].reshape(grid)[top:bot])
        if mode == "fixed":
            for i in range(sweeps):
                s.sweep(check=(i == sweeps - 1))
            s.reduce_delta()
        elif mode == "edit":
            for i in range(sweeps):
                s.sweep()
            v, types = _edits(grid, seed)
            free = s.set_cells(v, types)   # reported in place of the free-cell count: edits this rank owns
            for i in range(sweeps + 3):
                s.sweep(check=(i == sweeps + 2))
            s.reduce_delta()
        elif mode == "steps":   # the driver's own loop (pairs where it may), `sweeps` iterations, the last check's delta
            done = 0
            while done < sweeps:
                k, check = s.advance(sweeps - done)
                done += k
                if check:
                    s.reduce_delta()
        else:
            s.solve()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), u=s.owned(), lo=s.lo, hi=s.hi, delta=s.delta,
                 iteration=s.iteration, free=free)
    finally:
        dist.destroy_process_group()


def _edits(grid, seed):
    """A live-map update (epic_navigation_node_harmonic.cpp:357-380): new goals, new obstacles, freed cells -- placed on
    and around the slab seams of a 2- and a 3-way split, with a repeated cell (the later edit wins) and junk entries."""
    _, locked = synthetic_grid(grid, seed, 0.08)
    lk = locked.reshape(grid)
    rows, cols = grid
    v, t = [], []
    for r in sorted({rows // 2 - 1, rows // 2, rows // 2 + 1, rows // 3, rows // 3 + 1, 2 * rows // 3, 5, rows - 3}):
        v += [(7 + r % 5, r), (cols // 2, r), (cols - 4, r)]
        t += [1, 0, 1]
    blocked = np.argwhere(lk[1:-1, 1:-1] == 1)[::7] + 1          # some obstacles become free space
    v += [(int(c), int(r)) for r, c in blocked]
    t += [2] * len(blocked)
    v += [(cols // 2, rows // 2), (cols // 2, rows // 2)]        # same cell twice: obstacle, then goal
    t += [1, 0]
    v += [(cols + 5, 3), (3, rows + 9), (4, 4)]                  # out of range / unknown type: ignored
    t += [0, 1, 7]
    return np.array(v, dtype=np.uint32), np.array(t, dtype=np.uint32)


def _run(world, grid, seed, sweeps, mode, tmp_path, halo=8, scheme="jacobi", problem=None, pairs=False):
    mp.spawn(_worker, args=(world, _free_port(), grid, seed, sweeps, mode, str(tmp_path), halo, scheme, problem, pairs),
             nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    field = np.concatenate([p["u"] for p in parts], axis=0)
    return field, parts


def test_partition_and_row_generator():
    assert partition_rows(10, 3) == [(0, 4), (4, 7), (7, 10)]
    m = [37, 50]
    u, lk = synthetic_grid(m, 5, 0.08)
    for lo, hi in ((0, 37), (0, 12), (11, 26), (25, 37)):
        ur, lr = synthetic_rows(m, lo, hi, 5, 0.08)
        assert np.array_equal(ur, u.reshape(m)[lo:hi].ravel()) and np.array_equal(lr, lk.reshape(m)[lo:hi].ravel())


@pytest.mark.parametrize("world,halo", [(2, 1), (2, 8), (3, 1), (3, 3), (3, 5)])
def test_fixed_sweeps_equal_single_domain(world, halo, tmp_path):
    """halo = ghost depth G: G rows are traded every G sweeps (1 = a row every sweep).  25 sweeps is not a multiple of
    3, 5 or 8, so the run ends between two exchanges, on partly stale ghost rows but exact owned rows."""
    grid, seed, sweeps = [37, 50], 5, 25
    field, parts = _run(world, grid, seed, sweeps, "fixed", tmp_path, halo)
    u0, locked = synthetic_grid(grid, seed, 0.08)
    p = O.Problem(grid, u0, locked)
    assert O.oracle().oracle_jacobi_run(ct.byref(p.h), sweeps) == 0
    assert np.array_equal(field.ravel(), p.u)
    assert all(float(q["delta"]) == float(p.h.delta) for q in parts)
    assert sum(int(q["free"]) for q in parts) == int((locked.reshape(grid)[1:-1, 1:-1] == 0).sum())


def test_solve_equals_single_domain_jacobi(tmp_path):
    grid, seed = [30, 41], 9
    field, parts = _run(2, grid, seed, 0, "solve", tmp_path)
    u0, locked = synthetic_grid(grid, seed, 0.08)
    p = O.Problem(grid, u0, locked, 1e-6, 10)
    assert O.oracle().oracle_jacobi_complete(ct.byref(p.h)) == 0
    assert all(int(q["iteration"]) == int(p.h.currentIteration) for q in parts)
    assert np.array_equal(field.ravel(), p.u)


@pytest.mark.parametrize("world,halo,sweeps", [(2, 1, 11), (2, 8, 11), (3, 4, 10), (3, 5, 13)])
def test_set_cells_between_sweeps_equals_single_domain(world, halo, sweeps, tmp_path):
    """edits are routed to the slab(s) holding the cell (owned or ghost) and applied to both
    ping-pong buffers; the exchange cadence (edits land between, on and off an exchange sweep) must not matter."""
    grid, seed = [37, 50], 5
    field, parts = _run(world, grid, seed, sweeps, "edit", tmp_path, halo)
    u0, locked = synthetic_grid(grid, seed, 0.08)
    p = O.Problem(grid, u0, locked)
    lib = O.oracle()
    assert lib.oracle_jacobi_run(ct.byref(p.h), sweeps) == 0
    v, types = _edits(grid, seed)
    ok = types <= 2
    assert lib.oracle_set_cells_2d(ct.byref(p.h), int(ok.sum()), np.ascontiguousarray(v[ok]).ctypes.data_as(
        ct.POINTER(ct.c_uint)), np.ascontiguousarray(types[ok]).ctypes.data_as(ct.POINTER(ct.c_uint))) == 0
    assert lib.oracle_jacobi_run(ct.byref(p.h), sweeps + 3) == 0
    assert np.array_equal(field.ravel(), p.u)
    assert all(float(q["delta"]) == float(p.h.delta) for q in parts)
    in_range = (v[:, 0] < grid[1]) & (v[:, 1] < grid[0]) & ok
    assert sum(int(q["free"]) for q in parts) == int(in_range.sum())


@pytest.mark.parametrize("world,halo", [(2, 1), (2, 8), (3, 3), (3, 5)])
def test_redblack_half_sweeps_equal_single_domain(world, halo, tmp_path):
    """The reference's own scheme on slabs: 25 in-place half-sweeps (a count that ends between two exchanges) equal the
    checker's red-black half-sweeps on the whole grid, whatever the slab's first row parity (37 rows over 2 and 3 ranks
    give even and odd starts)."""
    grid, seed, sweeps = [37, 50], 5, 25
    field, parts = _run(world, grid, seed, sweeps, "fixed", tmp_path, halo, "redblack")
    u0, locked = synthetic_grid(grid, seed, 0.08)
    p = O.Problem(grid, u0, locked)
    lib = O.oracle()
    for i in range(sweeps):
        (lib.oracle_update_and_check if i == sweeps - 1 else lib.oracle_update)(ct.byref(p.h))
    assert np.array_equal(field.ravel(), p.u)
    assert all(float(q["delta"]) == float(p.h.delta) for q in parts)


@pytest.mark.parametrize("name,world", [("g2d_64", 2), ("g2d_23x37", 3), ("g2d_70x66_dense", 2)])
def test_redblack_slab_solve_is_the_reference_result(goldens, name, world, tmp_path):
    """Distributed solve with the reference's scheme against vectors the REFERENCE produced (tests/golden/small_grids.npz):
    same number of half-sweeps, same final delta, same field, bit for bit."""
    g, info = goldens["small"], goldens["manifest"]["small"][name]
    grid = [int(v) for v in g[name + "/m"]]
    problem = dict(u0=np.asarray(g[name + "/u0"], dtype=np.float32), locked=np.asarray(g[name + "/locked"], dtype=np.uint32),
                   stagger=int(info["stagger"]), epsilon=float(info["epsilon"]))
    field, parts = _run(world, grid, 0, 0, "solve", tmp_path, 4, "redblack", problem)
    assert all(int(q["iteration"]) == info["iterations"] for q in parts)
    assert all(float(q["delta"]) == info["delta"] for q in parts)
    assert np.array_equal(field.ravel(), g[name + "/converged"])


@pytest.mark.parametrize("world,halo", [(2, 8), (3, 3)])
def test_jacobi_solve_hands_over_where_plain_jacobi_never_ends(world, halo, tmp_path):
    """The nav_core plugin's second makePlan (tests/jacobi_handover_case.py): Jacobi's two colour chains stagnate one ulp
    apart there.  The slab driver takes the same decision as harmonic_execute_gpu and the checker -- on the all-reduced
    delta, so on every rank at the same iteration -- and ends in the checker's field after the checker's iteration count."""
    from jacobi_handover_case import GRID, two_goal_sequence, set_goal
    lib = O.oracle()
    u, locked, goals = two_goal_sequence()
    set_goal(u, locked, *goals[0])
    first = O.Problem(GRID, u, locked)
    assert lib.oracle_complete(ct.byref(first.h)) == 0
    u = first.u.reshape(GRID).copy()
    set_goal(u, locked, *goals[1])
    problem = dict(u0=u.ravel().copy(), locked=locked.ravel().copy(), stagger=100, epsilon=1e-6)
    field, parts = _run(world, GRID, 0, 0, "solve", tmp_path, halo, "jacobi", problem)
    p = O.Problem(GRID, u, locked)
    assert lib.oracle_jacobi_complete(ct.byref(p.h)) == 0
    assert all(int(q["iteration"]) == int(p.h.currentIteration) for q in parts)
    assert all(float(q["delta"]) == float(p.h.delta) < 1e-6 for q in parts)
    assert np.array_equal(field.ravel(), p.u)


@pytest.mark.parametrize("world,halo", [(1, 8), (2, 8), (2, 2), (2, 3), (3, 5), (3, 1)])
def test_pairs_of_iterations_as_one_pass_equal_single_domain(world, halo, tmp_path):
    """The driver runs two plain iterations as one backend pass (HIP: the fused double sweep of the tol math) wherever
    neither is a check and neither ends with an exchange -- a pass leaves TWO more ghost rows stale.  37 iterations at
    stagger 10 with G = 1, 2, 3, 5, 8: pairs, singles before an exchange, singles around every check; the result must be
    the single domain's, bit for bit, and so must the delta of the last check (iteration 30)."""
    grid, seed, sweeps = [37, 50], 5, 37
    field, parts = _run(world, grid, seed, sweeps, "steps", tmp_path, halo, pairs=True)
    u0, locked = synthetic_grid(grid, seed, 0.08)
    p = O.Problem(grid, u0, locked)
    lib = O.oracle()
    assert lib.oracle_jacobi_run(ct.byref(p.h), 31) == 0          # iterations 0..30: the 31st is the last check
    want_delta = float(p.h.delta)
    assert lib.oracle_jacobi_run(ct.byref(p.h), sweeps - 31) == 0
    assert np.array_equal(field.ravel(), p.u)
    assert all(int(q["iteration"]) == sweeps and float(q["delta"]) == want_delta for q in parts)


def test_solve_with_pairs_equals_single_domain_jacobi(tmp_path):
    grid, seed = [30, 41], 9
    field, parts = _run(2, grid, seed, 0, "solve", tmp_path, 4, pairs=True)
    u0, locked = synthetic_grid(grid, seed, 0.08)
    p = O.Problem(grid, u0, locked, 1e-6, 10)
    assert O.oracle().oracle_jacobi_complete(ct.byref(p.h)) == 0
    assert all(int(q["iteration"]) == int(p.h.currentIteration) for q in parts)
    assert np.array_equal(field.ravel(), p.u)


def test_jacobi_solve_with_reference_checks_is_the_reference_loop(tmp_path, monkeypatch):
    """EPIC_HIP_JACOBI_CHECKS=reference in the rank-per-slab driver (round 6; the library: tests/test_gpu_jacobi_reference_checks.py): every check iteration
    of the Jacobi run is the reference's half-sweep, so the two ranks' solve is harmonic_complete_cpu's loop -- oracle_complete -- bit for bit."""
    monkeypatch.setenv("EPIC_HIP_JACOBI_CHECKS", "reference")
    grid, seed = [30, 41], 9
    field, parts = _run(2, grid, seed, 0, "solve", tmp_path)
    u0, locked = synthetic_grid(grid, seed, 0.08)
    p = O.Problem(grid, u0, locked, 1e-6, 10)
    assert O.oracle().oracle_complete(ct.byref(p.h)) in (0, 1)
    assert all(int(q["iteration"]) == int(p.h.currentIteration) and float(q["delta"]) == float(p.h.delta) for q in parts)
    assert np.array_equal(field.ravel(), p.u)
