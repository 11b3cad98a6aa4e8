"""The N > 1 path on real hardware with what a 1-GPU box offers: two processes, both on cuda:0, HIP sweeps on each
slab, halo rows staged through the host over gloo (epic_amd/slab.py::_exchange_staged).  On a multi-GPU node the same
code runs one rank per GPU with RCCL send/recv.  The distributed field must equal the single-domain checker bit for bit."""
import ctypes as ct
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import _oracle as O
from epic_amd.synthetic import synthetic_grid

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _edits(grid):
    rows, cols = grid
    v = [(cols // 2, rows // 2 - 1), (9, rows // 2), (cols - 7, rows // 3), (200, rows // 3 + 1), (17, 2 * rows // 3),
         (300, rows // 2 + 3), (300, rows // 2 + 3)]
    t = [1, 0, 1, 0, 2, 1, 2]
    return np.array(v, dtype=np.uint32), np.array(t, dtype=np.uint32)


def _map_worker(rank, world, port, name, out_dir):
    """Red-black solve of a reference map on `world` ranks sharing cuda:0."""
    import torch
    import torch.distributed as dist

    from epic_amd.slab import SlabSolver

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, u0, locked = O.load_png_reference_rule(os.path.join(O.ROOT, "tests", "golden", "maps", name + ".png"))
        s = SlabSolver(m, rank, world, device=torch.device("cuda:0"), stagger=100, epsilon=1e-6, scheme="redblack",
                       rows_per_task=1)
        top, bot = s.lo - s.g_top, s.hi + s.g_bot
        s.load_rows(u0.reshape(m)[top:bot], locked.reshape(m)[top:bot])
        iterations = s.solve()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), u=s.owned(), delta=s.delta, iterations=iterations)
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, grid, seed, sweeps, out_dir, edit=False):
    import torch
    import torch.distributed as dist

    from epic_amd.slab import SlabSolver

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = SlabSolver(grid, rank, world, device=torch.device("cuda:0"), stagger=10)
        s.load_synthetic(seed=seed, density=0.06)
        if edit:
            for i in range(sweeps):
                s.sweep()
            s.set_cells(*_edits(grid))
        for i in range(sweeps):
            s.sweep(check=(i == sweeps - 1))
        delta = s.reduce_delta()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), u=s.owned(), delta=delta)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_and_three_ranks_on_one_gpu_equal_single_domain(world, tmp_path):
    grid, seed, sweeps = [211, 530], 12, 30
    mp.spawn(_worker, args=(world, _free_port(), grid, seed, sweeps, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    field = np.concatenate([p["u"] for p in parts], axis=0)
    u0, locked = synthetic_grid(grid, seed, 0.06)
    p = O.Problem(grid, u0, locked)
    assert O.oracle().oracle_jacobi_run(ct.byref(p.h), sweeps) == 0
    assert np.array_equal(field.ravel(), p.u)
    assert all(float(q["delta"]) == float(p.h.delta) for q in parts)


def test_set_cells_on_slabs_equals_single_domain(tmp_path):
    """Live-map edits on and around the slab seams, HIP mask repack on every rank that holds the cell."""
    world, grid, seed, sweeps = 2, [211, 530], 12, 21
    mp.spawn(_worker, args=(world, _free_port(), grid, seed, sweeps, str(tmp_path), True), nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    field = np.concatenate([p["u"] for p in parts], axis=0)
    u0, locked = synthetic_grid(grid, seed, 0.06)
    p = O.Problem(grid, u0, locked)
    lib = O.oracle()
    assert lib.oracle_jacobi_run(ct.byref(p.h), sweeps) == 0
    v, t = _edits(grid)
    assert lib.oracle_set_cells_2d(ct.byref(p.h), len(t), v.ctypes.data_as(ct.POINTER(ct.c_uint)),
                                   t.ctypes.data_as(ct.POINTER(ct.c_uint))) == 0
    assert lib.oracle_jacobi_run(ct.byref(p.h), sweeps) == 0
    assert np.array_equal(field.ravel(), p.u)
    assert all(float(q["delta"]) == float(p.h.delta) for q in parts)


def test_redblack_slab_solve_of_a_reference_map_is_the_reference_result(goldens, tmp_path):
    """BASELINE config 1's little brother (basic.png, 256 x 256) solved on two ranks with the reference's own scheme:
    the same 23 801 half-sweeps, the same final delta and the same field as harmonic_complete_cpu produced
    (tests/golden/maps_converged.npz), bit for bit."""
    name, world = "basic", 2
    mp.spawn(_map_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    run = goldens["manifest"]["maps"][name]["runs"]["1e-06"]
    assert all(int(q["iterations"]) == run["iterations"] and float(q["delta"]) == run["delta"] for q in parts)
    field = np.concatenate([p["u"] for p in parts], axis=0)
    assert np.array_equal(field.ravel(), goldens["maps"][name + "/converged_1e-06"])


def _tol_worker(rank, world, port, grid, seed, sweeps, halo, out_dir):
    """The tol arithmetic on slabs through the driver's own loop: fused double sweeps (epic_hip_sweep2_2d) between
    exchanges and checks, single sweeps on them."""
    import torch
    import torch.distributed as dist

    from epic_amd.slab import SlabSolver

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = SlabSolver(grid, rank, world, device=torch.device("cuda:0"), stagger=10, math="tol", halo=halo)
        assert s.backend.pairs
        s.load_synthetic(seed=seed, density=0.06)
        done, pairs = 0, 0
        while done < sweeps:
            k, check = s.advance(sweeps - done)
            done += k
            pairs += k // 2        # a stretch of k plain iterations runs k // 2 fused passes (one library call: epic_hip_sweeps_2d)
            if check:
                s.reduce_delta()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), u=s.owned(), delta=s.delta, pairs=pairs)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,halo", [(1, 8), (2, 8), (3, 4)])
def test_tol_slabs_with_fused_double_sweeps_equal_the_checker(world, halo, tmp_path):
    grid, seed, sweeps = [211, 530], 12, 37
    mp.spawn(_tol_worker, args=(world, _free_port(), grid, seed, sweeps, halo, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    field = np.concatenate([p["u"] for p in parts], axis=0)
    u0, locked = synthetic_grid(grid, seed, 0.06)
    lib = O.oracle()
    p = O.Problem(grid, u0, locked)
    assert lib.oracle_tol_run(ct.byref(p.h), 31, 0) == 0           # the last check is iteration 30
    want_delta = float(p.h.delta)
    assert lib.oracle_tol_run(ct.byref(p.h), sweeps - 31, 0) == 0
    assert np.array_equal(field.ravel(), p.u)
    assert all(float(q["delta"]) == want_delta for q in parts)
    assert all(int(q["pairs"]) >= 8 for q in parts)                 # the passes did run as pairs


def _rccl_worker(rank, world, port, grid, seed, sweeps, halo, math, out_dir):
    """One rank per GPU, RCCL send/recv of device rows: the branch of epic_amd/slab.py::_exchange a 1-GPU box never takes."""
    import torch
    import torch.distributed as dist

    from epic_amd.slab import SlabSolver

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        s = SlabSolver(grid, rank, world, device=torch.device("cuda", rank), stagger=10, math=math, halo=halo)
        s.load_synthetic(seed=seed, density=0.06)
        done = 0
        while done < sweeps:
            k, check = s.advance(sweeps - done)
            done += k
            if check:
                s.reduce_delta()
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), u=s.owned(), delta=s.delta)
    finally:
        dist.destroy_process_group()


def _gpus():
    import torch

    return torch.cuda.device_count()


@pytest.mark.multi_gpu
@pytest.mark.parametrize("halo,math", [(1, "precise"), (8, "precise"), (8, "tol"), (3, "tol")])
def test_rccl_slabs_one_rank_per_gpu_equal_single_domain(halo, math, tmp_path):
    """Runs wherever at least two GPUs are visible (the 1-GPU boxes of this project skip it; EPIC_TEST_MULTI_GPU=0 opts out): 2 .. 4
    ranks, one per device, halo rows over RCCL on the second stream while the interior is swept; 37 iterations at stagger
    10, so checks fall on and off exchange iterations.  Bit-identical to the single domain, delta of the last check
    included."""
    if os.environ.get("EPIC_TEST_MULTI_GPU", "1") == "0":
        pytest.skip("EPIC_TEST_MULTI_GPU=0")
    ndev = _gpus()
    if ndev < 2:
        pytest.skip("needs >= 2 GPUs")
    world = min(ndev, 4)
    grid, seed, sweeps = [421, 1030], 12, 37
    mp.spawn(_rccl_worker, args=(world, _free_port(), grid, seed, sweeps, halo, math, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    field = np.concatenate([p["u"] for p in parts], axis=0)
    u0, locked = synthetic_grid(grid, seed, 0.06)
    lib = O.oracle()
    p = O.Problem(grid, u0, locked)
    run = (lambda k: lib.oracle_tol_run(ct.byref(p.h), k, 0)) if math == "tol" else (lambda k: lib.oracle_jacobi_run(ct.byref(p.h), k))
    assert run(31) == 0
    want_delta = float(p.h.delta)
    assert run(sweeps - 31) == 0
    assert np.array_equal(field.ravel(), p.u)
    assert all(float(q["delta"]) == want_delta for q in parts)
