"""NEXT rows §8f-2 / §8f-4: streamline extraction and the legacy SOR exports, against vectors the reference produced
(tests/golden/generate_path_goldens.py) and, where it is available, against the compiled reference live."""
import ctypes as ct
import hashlib
import os

import numpy as np
import pytest

import _oracle as O
from epic_amd import epic_harmonic as eh
from epic_amd.harmonic import Harmonic

E = eh._epic
G = os.path.join(O.ROOT, "tests", "golden")
F, D, U = ct.c_float, ct.c_double, ct.c_uint
PF, PD, PU = ct.POINTER(F), ct.POINTER(D), ct.POINTER(U)


@pytest.fixture(scope="module")
def paths():
    return np.load(os.path.join(G, "paths.npz"))


def _field(goldens, name):
    m, _, locked = O.load_png_reference_rule(os.path.join(G, "maps", name + ".png"))
    h = Harmonic()
    h.set_grid(m, goldens["maps"][name + "/converged_1e-06"], locked)
    return h


@pytest.mark.parametrize("name", ["basic", "umass", "maze"])
def test_potential_and_gradient_match_reference(goldens, paths, name, capfd):
    h = _field(goldens, name)
    xs, ys = paths[name + "/probe_x"], paths[name + "/probe_y"]
    for i in range(len(xs)):
        v, a, b = F(0), F(0), F(0)
        rc = E.harmonic_compute_potential_2d_cpu(h, xs[i], ys[i], ct.byref(v))
        assert rc == paths[name + "/pot_rc"][i]
        assert np.float32(v.value).tobytes() == paths[name + "/pot"][i].tobytes()
        rc = E.harmonic_compute_gradient_2d_cpu(h, xs[i], ys[i], 0.5, ct.byref(a), ct.byref(b))
        assert rc == paths[name + "/grad_rc"][i]
        assert np.float32(a.value).tobytes() == paths[name + "/gx"][i].tobytes()
        assert np.float32(b.value).tobytes() == paths[name + "/gy"][i].tobytes()
    err = capfd.readouterr().err
    assert "Error[harmonic_compute_potential_2d_cpu]: Invalid location." in err
    assert "Error[harmonic_compute_gradient_2d_cpu]: Failed to compute potential values." in err


def _check_path(paths, key, pts):
    assert pts.size // 2 == int(paths[key + "_k"])
    assert np.array_equal(np.frombuffer(hashlib.sha256(pts.tobytes()).digest(), dtype=np.uint8), paths[key + "_sha256"])
    assert np.array_equal(pts[:16], paths[key + "_head"]) and np.array_equal(pts[-16:], paths[key + "_tail"])


@pytest.mark.parametrize("name", ["basic", "umass", "maze"])
def test_streamlines_match_reference(goldens, paths, name):
    h = _field(goldens, name)
    for j in range(6):
        sx, sy, step, cd = paths[f"{name}/path{j}_start"]
        k, raw = U(0), PF()
        rc = E.harmonic_compute_path_2d_cpu(h, sx, sy, step, cd, 1000000, ct.byref(k), ct.byref(raw))
        assert rc == int(paths[f"{name}/path{j}_rc"])
        if rc != 0:
            assert not raw
            continue
        pts = np.ctypeslib.as_array(raw, shape=(2 * k.value,)).copy()
        assert E.harmonic_free_path_cpu(ct.byref(raw)) == 0 and not raw
        _check_path(paths, f"{name}/path{j}", pts)
        # the streamline ends in a goal cell (locked, u == 0)
        ex, ey = int(pts[-2] + 0.5), int(pts[-1] + 0.5)
        assert h.locked_array()[ey, ex] == 1


def test_path_validation_codes(goldens):
    h = _field(goldens, "basic")
    k, raw = U(0), PF()
    assert E.harmonic_compute_path_2d_cpu(Harmonic(), 1.0, 1.0, 0.2, 0.4, 10, ct.byref(k), ct.byref(raw)) == 2
    assert E.harmonic_compute_path_2d_cpu(h, -5.0, 1.0, 0.2, 0.4, 10, ct.byref(k), ct.byref(raw)) == 10
    keep = (F * 2)()
    busy = ct.cast(keep, PF)                      # path must be NULL on entry (harmonic_path_cpu.cpp:160)
    assert E.harmonic_compute_path_2d_cpu(h, 100.0, 100.0, 0.2, 0.4, 10, ct.byref(k), ct.byref(busy)) == 2
    assert E.harmonic_free_path_cpu(ct.byref(raw)) == 0   # freeing NULL is fine


def test_legacy_sor_and_path_match_reference(paths):
    w, h = (int(v) for v in paths["legacy/w_h"])
    locked = paths["legacy/locked"].astype(np.uint32)
    for tag, ctype, dtype, fn in (("float", F, np.float32, E.harmonic_legacy_sor_2d_float_cpu),
                                  ("double", D, np.float64, E.harmonic_legacy_sor_2d_double_cpu),
                                  ("long_double", ct.c_longdouble, np.longdouble, E.harmonic_legacy_sor_2d_long_double_cpu)):
        u = paths["legacy/u0"].astype(dtype)
        it = U(0)
        assert fn(w, h, ctype(1e-3), ctype(1.5), locked.ctypes.data_as(PU), u.ctypes.data_as(ct.POINTER(ctype)),
                  ct.byref(it)) == 0
        assert it.value == int(paths[f"legacy/{tag}_iter"])
        assert np.array_equal(u.astype(np.float64), paths[f"legacy/{tag}_u"])
    ud = paths["legacy/double_u"].copy()
    for j in range(2):
        sx, sy = paths[f"legacy/path{j}_start"]
        k, raw = U(0), PD()
        rc = E.harmonic_legacy_compute_path_2d_cpu(w, h, locked.ctypes.data_as(PU), ud.ctypes.data_as(PD), sx, sy, 0.2, 0.4,
                                                   4000, 0, ct.byref(k), ct.byref(raw))
        assert rc == int(paths[f"legacy/path{j}_rc"]) == 0
        pts = np.ctypeslib.as_array(raw, shape=(2 * k.value,)).copy()
        assert E.harmonic_legacy_free_path_cpu(ct.byref(raw)) == 0
        _check_path(paths, f"legacy/path{j}", pts)
    v = D(0)
    assert E.harmonic_legacy_compute_potential_2d_cpu(w, h, locked.ctypes.data_as(PU), ud.ctypes.data_as(PD), 5.3, 5.1,
                                                      ct.byref(v)) == 0
    assert 0.0 <= v.value <= 1.0


def test_all_30_reference_symbols_exported():
    """`nm -D --defined-only libepic/lib/libepic.so` of the reference lists 30 harmonic_* functions."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", eh.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = sorted(l.split()[-1] for l in out.splitlines() if " T harmonic_" in l)
    assert len(names) == 30, names
    ref_so = "/root/reference/libepic/lib/libepic.so"
    if os.path.exists(ref_so):
        r = subprocess.run(["nm", "-D", "--defined-only", ref_so], capture_output=True, text=True, check=True).stdout
        ref_names = sorted(l.split()[-1] for l in r.splitlines() if " T harmonic_" in l)
        assert names == ref_names


def test_legacy_python_classes(paths):
    """HarmonicLegacy / HarmonicLegacyMap (reference python package, harmonic_legacy.py:34-95, harmonic_legacy_map.py:38-123)
    over the exported legacy functions: the SOR run and a streamline equal the reference-generated vectors."""
    from epic_amd.harmonic_legacy import HarmonicLegacy, HarmonicLegacyMap

    w, h = (int(v) for v in paths["legacy/w_h"])
    solver = HarmonicLegacy()
    solver.set_grid(paths["legacy/u0"].astype(np.float64).reshape(h, w), paths["legacy/locked"].reshape(h, w))
    wall, cpu = solver.solve(omega=1.5, epsilon=1e-3)      # the parameters of the golden run
    assert wall >= 0 and cpu >= 0 and solver.currentIteration == int(paths["legacy/double_iter"])
    assert np.array_equal(solver.u_array().ravel(), paths["legacy/double_u"])
    assert "omega" in str(solver)
    # the map flavour: load, relax, walk; the streamline ends on a goal pixel (255)
    m = HarmonicLegacyMap().load(os.path.join(G, "maps", "basic.png"))
    assert (m.w, m.h) == (256, 256) and m.locked_array()[0, 0] == 1
    assert set(np.unique(m.u_array())) <= {0.0, 1.0}
    m.solve(omega=1.5, epsilon=1e-10)
    free = np.argwhere((m.image != 0) & (m.image != 255))
    ends_on_goal = 0
    for y, x in free[:: max(1, len(free) // 12)][:12]:
        try:
            path = m._compute_streamline(float(x), float(y))
        except RuntimeError:
            continue
        ex, ey = path[-1]
        if np.isfinite(ex) and np.isfinite(ey) and m.image[int(ey + 0.5), int(ex + 0.5)] == 255:
            ends_on_goal += 1
    assert ends_on_goal >= 1
