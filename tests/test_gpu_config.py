"""The library reads its environment ONCE per context (epic_amd/csrc/driver_config.cpp: struct Config) and says which path a
context is on (epic_hip_config_dump).  Round 4's driver read ~30 EPIC_HIP_* variables at scattered points, some per batch; now a
variable changed on a live context has no effect until the caller announces it (epic_hip_config_reload)."""
import ctypes as ct
import os

import numpy as np
import pytest

import _oracle as O
from epic_amd import epic_harmonic as eh
from epic_amd.harmonic import Harmonic
from epic_amd.synthetic import synthetic_grid

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]

E = eh._epic
KNOBS = ("EPIC_HIP_MATH", "EPIC_HIP_SCHEME", "EPIC_HIP_TRACK", "EPIC_HIP_TILE", "EPIC_HIP_TILE_HALO", "EPIC_HIP_NO_GRAPH", "EPIC_HIP_NO_FUSE",
         "EPIC_HIP_FUSE_MIN_CELLS", "EPIC_HIP_DEVICES", "EPIC_HIP_HALO", "EPIC_HIP_SPIN_US", "EPIC_HIP_3D_PAIR")


@pytest.fixture
def clean_env():
    prev = {k: os.environ.get(k) for k in KNOBS}
    for k in KNOBS:
        os.environ.pop(k, None)
    yield
    for k, v in prev.items():
        os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


def start(m, seed=5):
    u0, locked = synthetic_grid(m, seed, 0.05)
    h = Harmonic()
    h.set_grid(m, u0, locked)
    h.epsilon = 1e-6
    for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
        assert fn(h) == 0, fn.__name__
    assert E.harmonic_initialize_gpu(h, 1024) == 0
    return h, u0, locked


def stop(h):
    for fn in (E.harmonic_uninitialize_gpu, E.harmonic_uninitialize_dimension_size_gpu, E.harmonic_uninitialize_potential_values_gpu,
               E.harmonic_uninitialize_locked_gpu):
        assert fn(h) == 0, fn.__name__


def test_dump_names_the_knobs_the_state_and_the_path(clean_env):
    assert eh.config_dump(Harmonic()) is None                       # no context: nothing to say
    h, _, _ = start([200, 300])
    d = eh.config_dump(h)
    assert d["config"]["math"] == 0 and d["config"]["scheme"] == "redblack" and d["config"]["tile"] is True     # the library's defaults
    assert d["state"]["n"] == 2 and d["state"]["rows"] == 200 and d["state"]["cols"] == 300 and d["state"]["pitch"] == 512
    assert d["state"]["scheme"] == "redblack" and d["state"]["tracking"] is False
    assert d["path"]["plain_batch"].startswith("LDS tiles") and d["path"]["tile_plan"]["iterations_per_launch"] == E.epic_hip_tile_iterations(h) > 0
    assert E.epic_hip_set_scheme(h, eh.SCHEME_JACOBI) == 0 and E.epic_hip_set_math_mode(h, eh.MATH_TOL) == 0
    d = eh.config_dump(h)
    assert d["config"]["scheme"] == "redblack" and d["state"]["scheme"] == "jacobi" and d["state"]["math"] == 4   # config: as read; state: in force
    stop(h)
    os.environ["EPIC_HIP_TILE"] = "0"
    os.environ["EPIC_HIP_MATH"] = "tol"
    h, _, _ = start([200, 300])
    d = eh.config_dump(h)
    assert d["config"]["tile"] is False and d["config"]["math"] == 4 and "hipGraph" in d["path"]["plain_batch"]
    stop(h)
    h, _, _ = start([9, 40, 300])
    assert "two planes per wave" in eh.config_dump(h)["path"]["plain_batch"]       # 3-D, tol: sweep3d_pair_kernel
    stop(h)


def test_a_variable_changed_on_a_live_context_waits_for_the_reload(clean_env):
    m = [150, 260]
    h, u0, locked = start(m)
    assert E.epic_hip_tile_iterations(h) > 0
    os.environ["EPIC_HIP_TILE"] = "0"
    assert E.epic_hip_tile_iterations(h) > 0                         # read once: nothing changes behind the caller's back
    assert E.epic_hip_config_reload(h) == 0
    assert E.epic_hip_tile_iterations(h) == 0 and eh.config_dump(h)["config"]["tile"] is False
    # a mode set through the API survives a reload unless ITS variable changed
    assert E.epic_hip_set_scheme(h, eh.SCHEME_JACOBI) == 0
    os.environ["EPIC_HIP_TILE_HALO"] = "5"
    os.environ.pop("EPIC_HIP_TILE")
    assert E.epic_hip_config_reload(h) == 0
    d = eh.config_dump(h)
    assert d["state"]["scheme"] == "jacobi" and d["path"]["tile_plan"]["iterations_per_launch"] == 5
    os.environ["EPIC_HIP_SCHEME"] = "redblack"                         # (the library default, but the variable CHANGED: absent -> given ... same value)
    os.environ["EPIC_HIP_MATH"] = "tol"
    assert E.epic_hip_config_reload(h) == 0
    d = eh.config_dump(h)
    assert d["state"]["math"] == 4 and d["state"]["scheme"] == "jacobi"    # math's variable changed; scheme's parsed value did not
    # results do not depend on any of it: 23 iterations against the checker's statement of tol Jacobi
    assert E.epic_hip_update_n_gpu(h, 23, 1) in (0, 1)
    assert E.harmonic_get_potential_values_gpu(h) == 0
    p = O.Problem(m, u0, locked)
    assert O.oracle().oracle_tol_run(ct.byref(p.h), 23, 0) == 0
    assert np.array_equal(h.u_array().ravel(), p.u) and float(h.delta) == float(p.h.delta)
    stop(h)
    assert E.epic_hip_config_reload(h) == eh.EPIC_ERROR_INVALID_DATA    # no context any more
    assert E.epic_hip_config_reload(None) == 0                          # the process-wide knobs (raw operators)


def test_slabs_report_their_seams_and_the_bounded_spin(clean_env):
    os.environ["EPIC_HIP_DEVICES"] = "0,0,0"
    os.environ["EPIC_HIP_HALO"] = "4"
    os.environ["EPIC_HIP_SPIN_US"] = "7"
    h, u0, locked = start([96, 300])
    d = eh.config_dump(h)
    assert d["state"]["slabs"] == 3 and d["path"]["halo"] == 4 and d["path"]["spin_us"] == 7 and d["path"]["issuing_threads"] is True
    assert [s["transport"] for s in d["path"]["seams"]] == ["same-device", "same-device"] and d["path"]["plain_batch"].startswith("slabs")
    assert E.epic_hip_set_scheme(h, eh.SCHEME_JACOBI) == 0
    assert E.epic_hip_update_n_gpu(h, 19, 1) in (0, 1)
    assert E.harmonic_get_potential_values_gpu(h) == 0
    p = O.Problem([96, 300], u0, locked)
    assert O.oracle().oracle_jacobi_run(ct.byref(p.h), 19) == 0
    assert np.array_equal(h.u_array().ravel(), p.u)
    stop(h)
