"""Randomised differential campaign: the library on the GPU against the checker (oracle/), tolerance 0.

Every case draws a grid (2-D up to ~0.8 Mcell or 3-D up to ~0.3 Mcell, ragged sizes, random obstacle density, a few goals, a
non-uniform start), an arithmetic and scheme (the library's default, precise Jacobi, tol red-black, tol Jacobi), a number of
iterations ending in a check, and a random setting of the knobs that select the code path (tiles on / off with random ring
depth, tile height and width; fused passes from 0 cells on; work lists forced; tracked pairs; graphs on / off; task heights) --
none of which may change a bit.  The field and delta after the iterations are compared with the checker's statement of the same
iterations (oracle_update* for the default, oracle_jacobi_run, oracle_tol_run).

    python tests/fuzz_gpu_parity.py [--cases 300] [--seed 1]         (test infrastructure: uses oracle/, like the tests)
tests/test_gpu_fuzz.py runs a short campaign of it under pytest.
"""
import argparse
import ctypes as ct
import json
import os
import sys

import numpy as np

os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))
os.environ.setdefault("EPIC_HIP_STUDY", "1")   # the knobs drawn below are study knobs: read only when asked for (epic_amd/csrc/driver_config.cpp)   # (the checker's OpenMP: the GPU box shows 256 cores and grants 16)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402
from epic_amd import epic_harmonic as eh  # noqa: E402
from epic_amd.harmonic import Harmonic  # noqa: E402
from epic_amd.synthetic import synthetic_grid  # noqa: E402

E = eh._epic
KNOBS = ("EPIC_HIP_DEFER", "EPIC_HIP_DEVICES", "EPIC_HIP_HALO", "EPIC_HIP_NO_PEER", "EPIC_HIP_THREADS", "EPIC_HIP_TILE", "EPIC_HIP_TILE_HALO", "EPIC_HIP_TILE_ROWS", "EPIC_HIP_TILE_WIDTH", "EPIC_HIP_TILE_PIPELINE", "EPIC_HIP_FUSE_MIN_CELLS",
         "EPIC_HIP_NO_FUSE", "EPIC_HIP_NO_GRAPH", "EPIC_HIP_FLAGS", "EPIC_HIP_TRACK", "EPIC_HIP_TRACK_PAIRS", "EPIC_HIP_TRACK_PAIR_ROWS", "EPIC_HIP_FUSED_ROWS",
         "EPIC_HIP_ROWS_PER_TASK", "EPIC_HIP_3D_PAIR", "EPIC_HIP_3D_MARCH", "EPIC_HIP_3D_PAIR_ROWS", "EPIC_HIP_TRACK_SWITCH", "EPIC_HIP_MATH",
         "EPIC_HIP_SCHEME")
MODES = [("default", eh.MATH_PRECISE, eh.SCHEME_REDBLACK), ("precise jacobi", eh.MATH_PRECISE, eh.SCHEME_JACOBI),
         ("tol redblack", eh.MATH_TOL, eh.SCHEME_REDBLACK), ("tol jacobi", eh.MATH_TOL, eh.SCHEME_JACOBI)]


def draw_case(rng):
    big = rng.random() < 0.08    # now and then a grid of the size where the library changes gear (4 Mcell: fused passes, work lists)
    if big and rng.random() < 0.4:
        m = [int(rng.integers(60, 150)), int(rng.integers(60, 150)), int(rng.integers(250, 700))]
    elif big:
        m = [int(rng.integers(1500, 2600)), int(rng.integers(1800, 3000))]
    elif rng.random() < 0.3:
        m = [int(rng.integers(3, 70)), int(rng.integers(3, 70)), int(rng.integers(3, 300))]
        while np.prod(m) > 300000:
            m[int(rng.integers(0, 3))] //= 2
            m = [max(3, x) for x in m]
    else:
        m = [int(rng.integers(3, 900)), int(rng.integers(3, 1300))]
        if rng.random() < 0.2:
            m[int(rng.integers(0, 2))] = int(rng.integers(3, 12))     # thin grids
    dens = float(rng.choice([0.0, 0.02, 0.05, 0.15, 0.35]))
    seed = int(rng.integers(1, 1 << 30))
    u0, locked = synthetic_grid(m, seed, dens)
    free = np.flatnonzero(locked == 0)
    for idx in rng.choice(free, size=min(free.size, int(rng.integers(0, 4))), replace=False) if free.size else []:
        u0[idx] = 0.0
        locked[idx] = 1
    if rng.random() < 0.8 and free.size > 8:
        O.scramble_free(m, u0, locked, seed=seed + 7, lo=float(rng.choice([-3.0, -40.0, -900.0])), hi=0.0)
    mode = MODES[int(rng.integers(0, len(MODES)))]
    k = int(rng.integers(1, 14 if big else 45))
    env = {}
    if len(m) == 2:
        env["EPIC_HIP_TILE"] = rng.choice(["0", "1", None])
        if env["EPIC_HIP_TILE"] != "0" and rng.random() < 0.6:
            env["EPIC_HIP_TILE_HALO"] = str(int(rng.integers(1, 28)))
            if rng.random() < 0.5:
                env["EPIC_HIP_TILE_ROWS"] = str(2 * int(rng.integers(2, 30)))
            env["EPIC_HIP_TILE_WIDTH"] = rng.choice(["64", "128", None])
        env["EPIC_HIP_TILE_PIPELINE"] = rng.choice(["0", None])
        if rng.random() < 0.4:
            env["EPIC_HIP_FUSE_MIN_CELLS"] = "0"
            env["EPIC_HIP_TILE"] = "0"
            if rng.random() < 0.5:
                env["EPIC_HIP_FUSED_ROWS"] = str(int(rng.integers(4, 60)))
            env["EPIC_HIP_FLAGS"] = rng.choice(["3", "6", None])   # (bit 2: the fused passes' chunk tightening; bit 0: march direction)
        env["EPIC_HIP_TRACK_PAIRS"] = rng.choice(["0", None])
        if rng.random() < 0.3:
            env["EPIC_HIP_TRACK_PAIR_ROWS"] = str(int(rng.integers(2, 40)))
        if rng.random() < 0.3:
            env["EPIC_HIP_ROWS_PER_TASK"] = str(int(rng.integers(1, 40)))
    else:
        env["EPIC_HIP_3D_PAIR"] = rng.choice(["0", None])
        env["EPIC_HIP_3D_MARCH"] = rng.choice(["x0", None])
        if rng.random() < 0.4:
            env["EPIC_HIP_3D_PAIR_ROWS"] = str(int(rng.integers(4, 50)))
    env["EPIC_HIP_NO_GRAPH"] = rng.choice(["1", None])
    env["EPIC_HIP_TRACK"] = rng.choice(["0", "1", None])
    env["EPIC_HIP_TRACK_SWITCH"] = rng.choice(["0", "2", None])
    if m[0] >= 48 and rng.random() < 0.25:   # the in-library slabs (the device repeated: one GPU suffices), 2-4 of them
        env["EPIC_HIP_DEVICES"] = ",".join(["0"] * int(rng.integers(2, 5)))
        env["EPIC_HIP_HALO"] = rng.choice([None, "1", "2", "3", "5"])
        env["EPIC_HIP_NO_PEER"] = rng.choice(["1", None])
        env["EPIC_HIP_THREADS"] = rng.choice(["0", None])
    env = {a: (None if b is None else str(b)) for a, b in env.items()}
    # live edits of the resident state between the two calls (the navigation node's set_cells: 0 = goal, 1 = obstacle, 2 = free; cells
    # out of range and unknown types are skipped by the reference): 2-D, one device or slabs
    edits = None
    if len(m) == 2 and k > 3 and rng.random() < 0.3:
        n = int(rng.integers(1, 12))
        v = np.stack([rng.integers(0, m[1] + 2, n), rng.integers(0, m[0] + 2, n)], axis=1).astype(np.uint32)   # (x = column, y = row)
        edits = (np.ascontiguousarray(v), rng.integers(0, 4, n).astype(np.uint32))
    return m, u0, locked, mode, k, env, edits


def first_part(k):
    return k // 2 if k > 3 else 0


def checker(m, u0, locked, mode, k, edits=None):
    p = O.Problem(m, u0, locked)
    name, math, scheme = mode
    lib = O.oracle()

    def run(n, check):
        if n == 0:
            return
        if math == eh.MATH_TOL:
            assert lib.oracle_tol_run(ct.byref(p.h), n, 1 if scheme == eh.SCHEME_REDBLACK else 0) == 0
        elif scheme == eh.SCHEME_JACOBI:
            assert lib.oracle_jacobi_run(ct.byref(p.h), n) == 0
        else:
            for i in range(n):
                (lib.oracle_update_and_check if check and i == n - 1 else lib.oracle_update)(ct.byref(p.h))

    first = first_part(k)
    run(first, False)
    if edits is not None and first:
        v, t = edits
        assert lib.oracle_set_cells_2d(ct.byref(p.h), len(t), v.ctypes.data_as(eh._UP), t.ctypes.data_as(eh._UP)) == 0
    run(k - first, True)
    return p.u.copy(), float(p.h.delta)


LAST = {"dump": None}   # epic_hip_config_dump of the latest library() context (which knobs it read, which kernel path it took)


def library(m, u0, locked, mode, k, env, edits=None):
    prev = {a: os.environ.get(a) for a in KNOBS}
    for a in KNOBS:
        os.environ.pop(a, None)
    for a, b in env.items():
        if b is not None:
            os.environ[a] = b
    assert E.epic_hip_config_reload(None) == 0   # EPIC_HIP_FLAGS is the process's
    try:
        h = Harmonic()
        h.set_grid(m, u0, locked)
        h.epsilon = 1e-6
        for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
            assert fn(h) == 0
        assert E.harmonic_initialize_gpu(h, 1024) == 0
        assert E.epic_hip_set_math_mode(h, mode[1]) == 0 and E.epic_hip_set_scheme(h, mode[2]) == 0
        # in two calls at a random split, so that batches, graphs and lists are entered and left mid-way
        first = first_part(k)
        if first:
            assert E.epic_hip_update_n_gpu(h, first, 0) == 0
            if edits is not None:
                v, t = edits
                assert E.harmonic_utilities_set_cells_2d_gpu(h, 1024, len(t), v.ctypes.data_as(eh._UP), t.ctypes.data_as(eh._UP)) == 0
        assert E.epic_hip_update_n_gpu(h, k - first, 1) in (0, 1)
        assert h.currentIteration == k
        assert E.harmonic_get_potential_values_gpu(h) == 0
        LAST["dump"] = eh.config_dump(h)   # the library's own account of this context: printed with a mismatch
        delta = float(h.delta)
        for fn in (E.harmonic_uninitialize_gpu, E.harmonic_uninitialize_dimension_size_gpu, E.harmonic_uninitialize_potential_values_gpu,
                   E.harmonic_uninitialize_locked_gpu):
            assert fn(h) == 0
        return h.u_array().ravel().copy(), delta
    finally:
        for a, b in prev.items():
            os.environ.pop(a, None) if b is None else os.environ.__setitem__(a, b)
        E.epic_hip_config_reload(None)


def campaign(cases, seed, verbose=True):
    rng = np.random.default_rng(seed)
    bad = []
    for i in range(cases):
        m, u0, locked, mode, k, env, edits = draw_case(rng)
        want, wdelta = checker(m, u0, locked, mode, k, edits)
        got, gdelta = library(m, u0, locked, mode, k, env, edits)
        ok = np.array_equal(got, want) and gdelta == wdelta
        if verbose or not ok:
            print(f"case {i:4d} {'ok  ' if ok else 'FAIL'} {str(m):18s} {mode[0]:15s} k={k:2d} " + " ".join(f"{a[9:]}={b}" for a, b in env.items() if b is not None)
                  + (f" edits={len(edits[1])}" if edits is not None else ""), flush=True)
        if not ok:
            diff = np.flatnonzero(got != want)
            bad.append(dict(case=i, seed=seed, m=m, mode=mode[0], k=k, env=env, cells=int(diff.size), first=int(diff[0]) if diff.size else -1,
                            delta=(gdelta, wdelta), library=LAST["dump"]))
            print("    the library's account of that context:", json.dumps(LAST["dump"]), flush=True)
    return bad


# ---- the fine-grained API (round 6: harmonic_update_gpu counts, the library enqueues whole blocks) -----------------------------------
def draw_script(rng, m, big):
    """A caller's script over the fine-grained entry points, as the navigation node mixes them (src/epic_navigation_node_harmonic.cpp:
    165-189 ticks, :357-380 setCells, :614-626 the read-back of srvComputePath): 'u' harmonic_update_gpu, 'c' harmonic_update_and_check_gpu,
    ('n', k, check) epic_hip_update_n_gpu, ('e', v, types) set_cells, 'r' get_potential_values (compared with the checker there and then),
    ('i', it) the caller sets currentIteration itself."""
    ops = []
    budget = int(rng.integers(4, 24 if big else 90))
    while budget > 0:
        r = rng.random()
        if r < 0.45:      # a tick of the node: one check, then steps - 1 plain updates
            steps = int(rng.integers(1, min(budget, 40) + 1))
            ops.append("c")
            ops.extend("u" * (steps - 1))
            budget -= steps
        elif r < 0.60:    # plain updates on their own
            k = int(rng.integers(1, min(budget, 25) + 1))
            ops.extend("u" * k)
            budget -= k
        elif r < 0.70:
            k = int(rng.integers(1, min(budget, 30) + 1))
            ops.append(("n", k, int(rng.integers(0, 2))))
            budget -= k
        elif r < 0.82 and len(m) == 2:
            n = int(rng.integers(1, 9))
            v = np.stack([rng.integers(0, m[1] + 2, n), rng.integers(0, m[0] + 2, n)], axis=1).astype(np.uint32)
            t = rng.integers(0, 4, n).astype(np.uint32)
            # (one edit per cell and call: the reference's device scatter is one thread per edit -- harmonic_utilities_gpu.cu:38-64 --, two
            #  edits of one cell in one call race there as here; its CPU form applies them in order)
            _, keep = np.unique(v[:, 1].astype(np.int64) * (m[1] + 2) + v[:, 0], return_index=True)
            keep.sort()
            ops.append(("e", np.ascontiguousarray(v[keep]), np.ascontiguousarray(t[keep])))
        elif r < 0.94:
            ops.append("r")
        else:
            ops.append(("i", int(rng.integers(0, 1000))))
    ops.append("c")
    ops.append("r")
    return ops


def node_case(rng):
    m, u0, locked, mode, k, env, edits = draw_case(rng)
    big = int(np.prod(m)) > 3000000
    env["EPIC_HIP_DEFER"] = rng.choice(["0", None, None, None])
    return m, u0, locked, mode, env, draw_script(rng, m, big)


REF_CHECKS = False   # --ref-checks: EPIC_HIP_JACOBI_CHECKS=reference on the device, stated here for the scripts and by the checkers' loops for --complete


def checker_script(m, u0, locked, mode, ops):
    """The checker's statement of the script: (field, delta) at every 'r', in order."""
    p = O.Problem(m, u0, locked)
    name, math, scheme = mode
    lib = O.oracle()
    pending = [0]
    delta = [0.0]

    def run(ends_with_check):
        n = pending[0]
        pending[0] = 0
        if n == 0:
            return
        if REF_CHECKS and ends_with_check and scheme == eh.SCHEME_JACOBI and len(m) != 4:
            # EPIC_HIP_JACOBI_CHECKS=reference: the check is the reference's half-sweep of that iteration's colour, in place
            if n > 1:
                assert (lib.oracle_tol_run(ct.byref(p.h), n - 1, 0) if math == eh.MATH_TOL else lib.oracle_jacobi_run(ct.byref(p.h), n - 1)) == 0
            if math == eh.MATH_TOL:
                assert lib.oracle_tol_run(ct.byref(p.h), 1, 1) == 0
            else:
                assert lib.oracle_update_and_check(ct.byref(p.h)) in (0, 1)
            delta[0] = float(p.h.delta)
            return
        if math == eh.MATH_TOL:
            assert lib.oracle_tol_run(ct.byref(p.h), n, 1 if scheme == eh.SCHEME_REDBLACK else 0) == 0
        elif scheme == eh.SCHEME_JACOBI:
            assert lib.oracle_jacobi_run(ct.byref(p.h), n) == 0
        else:
            for i in range(n):
                (lib.oracle_update_and_check if ends_with_check and i == n - 1 else lib.oracle_update)(ct.byref(p.h))
        if ends_with_check:
            delta[0] = float(p.h.delta)

    shots = []
    for op in ops:
        if op == "u":
            pending[0] += 1
        elif op == "c":
            pending[0] += 1
            run(True)
        elif op == "r":
            run(False)
            shots.append((p.u.copy(), delta[0]))
        elif op[0] == "n":
            pending[0] += op[1]
            run(bool(op[2]))
        elif op[0] == "e":
            run(False)
            assert lib.oracle_set_cells_2d(ct.byref(p.h), len(op[2]), op[1].ctypes.data_as(eh._UP), op[2].ctypes.data_as(eh._UP)) == 0
        elif op[0] == "i":
            run(False)
            p.h.currentIteration = op[1]
    return shots


def library_script(m, u0, locked, mode, env, ops):
    prev = {a: os.environ.get(a) for a in KNOBS}
    for a in KNOBS:
        os.environ.pop(a, None)
    for a, b in env.items():
        if b is not None:
            os.environ[a] = b
    assert E.epic_hip_config_reload(None) == 0
    try:
        h = Harmonic()
        h.set_grid(m, u0, locked)
        h.epsilon = 1e-6
        for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
            assert fn(h) == 0
        assert E.harmonic_initialize_gpu(h, 1024) == 0
        assert E.epic_hip_set_math_mode(h, mode[1]) == 0 and E.epic_hip_set_scheme(h, mode[2]) == 0
        shots = []
        delta = 0.0
        for op in ops:
            if op == "u":
                assert E.harmonic_update_gpu(h, 1024) == 0
            elif op == "c":
                assert E.harmonic_update_and_check_gpu(h, 1024) in (0, 1)
                delta = float(h.delta)
            elif op == "r":
                assert E.harmonic_get_potential_values_gpu(h) == 0
                shots.append((h.u_array().ravel().copy(), delta))
            elif op[0] == "n":
                assert E.epic_hip_update_n_gpu(h, op[1], op[2]) in (0, 1)
                if op[2]:
                    delta = float(h.delta)
            elif op[0] == "e":
                assert E.harmonic_utilities_set_cells_2d_gpu(h, 1024, len(op[2]), op[1].ctypes.data_as(eh._UP), op[2].ctypes.data_as(eh._UP)) == 0
            elif op[0] == "i":
                h.currentIteration = op[1]
        LAST["dump"] = eh.config_dump(h)
        for fn in (E.harmonic_uninitialize_gpu, E.harmonic_uninitialize_dimension_size_gpu, E.harmonic_uninitialize_potential_values_gpu,
                   E.harmonic_uninitialize_locked_gpu):
            assert fn(h) == 0
        return shots
    finally:
        for a, b in prev.items():
            os.environ.pop(a, None) if b is None else os.environ.__setitem__(a, b)
        E.epic_hip_config_reload(None)


def script_text(ops):
    out, run = [], 0
    for op in list(ops) + [None]:
        if op == "u":
            run += 1
            continue
        if run:
            out.append("u*%d" % run)
            run = 0
        if op is None:
            break
        out.append(op if isinstance(op, str) else "%s%s" % (op[0], "" if op[0] == "e" else op[1]))
    return " ".join(out)


def campaign_node(cases, seed, verbose=True):
    """Random scripts of single calls, batches, edits, read-backs and caller-set iteration numbers: every read-back must show the checker's
    field bit for bit, every check its delta -- whatever the library deferred and however it enqueued it."""
    rng = np.random.default_rng(seed)
    bad = []
    for i in range(cases):
        m, u0, locked, mode, env, ops = node_case(rng)
        want = checker_script(m, u0, locked, mode, ops)
        got = library_script(m, u0, locked, mode, env, ops)
        ok = len(want) == len(got) and all(np.array_equal(g[0], w[0]) and g[1] == w[1] for g, w in zip(got, want))
        if verbose or not ok:
            print(f"node {i:4d} {'ok  ' if ok else 'FAIL'} {str(m):18s} {mode[0]:15s} " + " ".join(f"{a[9:]}={b}" for a, b in env.items() if b is not None)
                  + " | " + script_text(ops), flush=True)
        if not ok:
            first = next((j for j, (g, w) in enumerate(zip(got, want)) if not (np.array_equal(g[0], w[0]) and g[1] == w[1])), -1)
            bad.append(dict(case=i, seed=seed, m=m, mode=mode[0], env=env, script=script_text(ops), first_bad_readback=first,
                            cells=int((got[first][0] != want[first][0]).sum()) if first >= 0 else -1,
                            delta=(got[first][1], want[first][1]) if first >= 0 else None, library=LAST["dump"]))
            print("    the library's account of that context:", json.dumps(LAST["dump"]), flush=True)
    return bad


def draw_complete(rng):
    """A whole relaxation: a small grid (the checker must finish it), a random epsilon and check interval."""
    if rng.random() < 0.25:
        m = [int(rng.integers(3, 22)), int(rng.integers(3, 22)), int(rng.integers(3, 40))]
    else:
        m = [int(rng.integers(3, 110)), int(rng.integers(3, 140))]
    dens = float(rng.choice([0.0, 0.05, 0.15, 0.3]))
    seed = int(rng.integers(1, 1 << 30))
    u0, locked = synthetic_grid(m, seed, dens)
    free = np.flatnonzero(locked == 0)
    for idx in rng.choice(free, size=min(free.size, int(rng.integers(0, 3))), replace=False) if free.size else []:
        u0[idx] = 0.0
        locked[idx] = 1
    mode = MODES[int(rng.integers(0, len(MODES)))]
    eps = float(rng.choice([1e-2, 1e-3, 1e-4, 1e-6]))
    stagger = int(rng.choice([1, 7, 50, 100, 128]))
    env = {}
    if len(m) == 2:
        env["EPIC_HIP_TILE"] = rng.choice(["0", "1", None])
        if env["EPIC_HIP_TILE"] != "0" and rng.random() < 0.5:
            env["EPIC_HIP_TILE_HALO"] = str(int(rng.integers(1, 20)))
            env["EPIC_HIP_TILE_WIDTH"] = rng.choice(["64", "128", None])
        env["EPIC_HIP_TILE_PIPELINE"] = rng.choice(["0", None])
        if rng.random() < 0.3:
            env["EPIC_HIP_FUSE_MIN_CELLS"] = "0"
            env["EPIC_HIP_TILE"] = "0"
        env["EPIC_HIP_TRACK_PAIRS"] = rng.choice(["0", None])
    else:
        env["EPIC_HIP_3D_PAIR"] = rng.choice(["0", None])
    env["EPIC_HIP_NO_GRAPH"] = rng.choice(["1", None])
    env["EPIC_HIP_TRACK"] = rng.choice(["0", "1", None])
    env["EPIC_HIP_TRACK_SWITCH"] = rng.choice(["0", "2", None])
    return m, u0, locked, mode, eps, stagger, {a: (None if b is None else str(b)) for a, b in env.items()}


def campaign_complete(cases, seed, verbose=True):
    """harmonic_complete_gpu against the checker's loops: field, iteration count and delta, tolerance 0 -- the library's default
    against oracle_complete (= harmonic_complete_cpu), precise Jacobi against oracle_jacobi_complete (with its hand-over to
    red-black), tol against oracle_tol_complete (with its finishing iterations)."""
    rng = np.random.default_rng(seed)
    bad = []
    names = {eh.MATH_PRECISE: "precise", eh.MATH_TOL: "tol"}
    for i in range(cases):
        m, u0, locked, mode, eps, stagger, env = draw_complete(rng)
        p = O.Problem(m, u0, locked, epsilon=eps, stagger=stagger)
        if mode[1] == eh.MATH_TOL:
            rc = O.oracle().oracle_tol_complete(ct.byref(p.h), 1 if mode[2] == eh.SCHEME_REDBLACK else 0)
        elif mode[2] == eh.SCHEME_JACOBI:
            rc = O.oracle().oracle_jacobi_complete(ct.byref(p.h))
        else:
            rc = O.oracle().oracle_complete(ct.byref(p.h))
        assert rc == 0, rc
        prev = {a: os.environ.get(a) for a in KNOBS}
        for a in KNOBS:
            os.environ.pop(a, None)
        for a, b in env.items():
            if b is not None:
                os.environ[a] = b
        os.environ["EPIC_HIP_MATH"] = names[mode[1]]
        os.environ["EPIC_HIP_SCHEME"] = "redblack" if mode[2] == eh.SCHEME_REDBLACK else "jacobi"
        assert E.epic_hip_config_reload(None) == 0
        try:
            h = Harmonic()
            h.set_grid(m, u0, locked)
            h.epsilon = eps
            h.numIterationsToStaggerCheck = stagger
            assert E.harmonic_complete_gpu(h, 1024) == 0
            got, its, delta = h.u_array().ravel().copy(), int(h.currentIteration), float(h.delta)
        finally:
            for a, b in prev.items():
                os.environ.pop(a, None) if b is None else os.environ.__setitem__(a, b)
            E.epic_hip_config_reload(None)
        ok = np.array_equal(got, p.u) and its == int(p.h.currentIteration) and delta == float(p.h.delta)
        if verbose or not ok:
            print(f"complete {i:4d} {'ok  ' if ok else 'FAIL'} {str(m):14s} {mode[0]:15s} eps {eps:g} stagger {stagger:3d} iterations {its} (checker {int(p.h.currentIteration)}) "
                  + " ".join(f"{a[9:]}={b}" for a, b in env.items() if b is not None), flush=True)
        if not ok:
            bad.append(dict(case=i, seed=seed, m=m, mode=mode[0], eps=eps, stagger=stagger, env=env, iterations=(its, int(p.h.currentIteration)),
                            cells=int((got != p.u).sum()), delta=(delta, float(p.h.delta))))
    return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--complete", action="store_true", help="whole relaxations (harmonic_complete_gpu) instead of fixed iteration counts")
    ap.add_argument("--node", action="store_true", help="scripts over the fine-grained entry points (single updates, checks, edits, read-backs)")
    ap.add_argument("--ref-checks", action="store_true", help="with --complete or --node: EPIC_HIP_JACOBI_CHECKS=reference on the device and in the checkers' statements")
    a = ap.parse_args()
    if a.ref_checks:
        assert a.complete or a.node, "--ref-checks goes with --complete (the checkers' whole-relaxation loops state it) or --node (checker_script states it)"
        REF_CHECKS = True
        os.environ["EPIC_HIP_JACOBI_CHECKS"] = "reference"
        O.oracle().oracle_set_jacobi_ref_checks.argtypes = (ct.c_int,)
        O.oracle().oracle_set_jacobi_ref_checks.restype = None
        O.oracle().oracle_set_jacobi_ref_checks(1)
    bad = (campaign_node if a.node else campaign_complete if a.complete else campaign)(a.cases, a.seed, verbose=not a.quiet)
    print(f"{a.cases} cases, seed {a.seed}: {len(bad)} mismatches")
    for b in bad:
        print(b)
    sys.exit(1 if bad else 0)
