"""Parity of the BENCHMARKED arithmetic (tol math, Jacobi) with the reference at the sizes the benchmark runs.

bench.py times BASELINE configs[2]: the synthetic 8192 x 8192 grid (seed 20240601).  The reference's harmonic_complete_cpu needs
~15 h for it, so this file pins the tol mode on that grid FAMILY in two steps:

  * 512 x 512 and 1024 x 1024 of the same generator and seed against fields the REFERENCE itself converged
    (tests/golden/synthetic_converged.npz, written by tests/golden/generate_synthetic_goldens.py from oracle/_ref): every mode
    within 1e-5 max(1, |u|), unreached cells equal, and `precise` + `redblack` -- the library default -- bit-identical to the
    reference including the iteration count and the final delta;
  * 8192 x 8192 on the device itself: `precise` + `redblack`, which the step above (and tests/test_gpu_parity.py on the
    reference's maps) shows to BE the reference's iteration, is the yard-stick for `tol` + `jacobi` at full size.

The error of a converged field grows with the domain's radius, which is why the small seeded grids of
tests/test_gpu_tol.py are not enough for the benchmark's claim.  Measured values are printed and recorded as test properties;
bench.py reports the same numbers in its `parity` object.

Reference behaviour matched: harmonic_complete_cpu (libepic/src/harmonic/harmonic_cpu.cpp:60-70 rounding sequence, :136-184 driver).
"""
import json
import os

import numpy as np
import pytest

import _oracle as O
from conftest import scheme_env
from epic_amd import epic_harmonic as eh
from epic_amd.harmonic import Harmonic
from epic_amd.synthetic import synthetic_grid

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

E = eh._epic
BAR = 1e-5
SEED = np.float32(-1e6)


@pytest.fixture(scope="module")
def synth_goldens():
    g = os.path.join(O.ROOT, "tests", "golden")
    return np.load(os.path.join(g, "synthetic_converged.npz")), json.load(open(os.path.join(g, "manifest.json")))["synthetic"]["grids"]


def relax(m, u0, locked, math, scheme, track=2):
    """harmonic_execute_gpu to eps = 1e-6 / stagger 100; returns (field, iterations, delta)."""
    h = Harmonic()
    h.set_grid(m, u0, locked)
    h.epsilon = 1e-6
    h.numIterationsToStaggerCheck = 100
    for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu,
               E.harmonic_initialize_locked_gpu):
        assert fn(h) == 0, fn.__name__
    assert E.epic_hip_set_math_mode(h, math) == 0 and E.epic_hip_set_scheme(h, scheme) == 0
    assert E.epic_hip_set_activity_tracking(h, track) == 0
    assert E.harmonic_execute_gpu(h, 1024) == 0
    for fn in (E.harmonic_uninitialize_dimension_size_gpu, E.harmonic_uninitialize_potential_values_gpu,
               E.harmonic_uninitialize_locked_gpu):
        assert fn(h) == 0, fn.__name__
    return h.u_array().ravel(), int(h.currentIteration), float(h.delta)


def distance(got, want, locked):
    """(max relative, max absolute) over reached free cells; unreached free cells and locked cells must be equal."""
    got, want, locked = np.ravel(got), np.ravel(want), np.ravel(locked)
    free = locked == 0
    reached = free & (want > -9e5)
    assert np.array_equal(got[~reached], want[~reached]), "locked cells and cells the front never reaches must be equal"
    d = np.abs(got[reached].astype(np.float64) - want[reached])
    rel = d / np.maximum(1.0, np.abs(want[reached]))
    return float(rel.max()), float(d.max())


MODES = [("precise", eh.MATH_PRECISE, "redblack", eh.SCHEME_REDBLACK), ("precise", eh.MATH_PRECISE, "jacobi", eh.SCHEME_JACOBI),
         ("tol", eh.MATH_TOL, "jacobi", eh.SCHEME_JACOBI), ("tol", eh.MATH_TOL, "redblack", eh.SCHEME_REDBLACK)]


@pytest.mark.parametrize("n", [512, 1024])
@pytest.mark.parametrize("mname,math,sname,scheme", MODES)
def test_synthetic_family_vs_reference_converged(synth_goldens, n, mname, math, sname, scheme, record_property):
    fields, info = synth_goldens
    info = info[str(n)]
    m = [n, n]
    u0, locked = synthetic_grid(m)     # seed 20240601, 5 % obstacles: the benchmark's generator
    want = fields["s%d/converged" % n]
    got, its, delta = relax(m, u0, locked, math, scheme)
    rel, ab = distance(got, want, locked)
    record_property("max_rel", rel)
    record_property("max_abs", ab)
    print("synthetic %d^2 %s %s: %d iterations (reference %d), delta %.3e, max rel %.3e, max abs %.3e"
          % (n, mname, sname, its, info["iterations"], delta, rel, ab))
    assert delta < 1e-6
    assert rel <= BAR, "converged field further than 1e-5 max(1, |u|) from the reference's"
    # (tol: the relaxation finishes with the reference's own iteration from delta < 10 eps on; where the tol phase freezes
    #  between two checks -- the front arrives and everything stops -- that phase starts from a field that already looks
    #  converged and walks the dead band for a few hundred iterations of its own: 4 501 against 3 801 at 512^2)
    assert abs(its - info["iterations"]) <= (0.02 if mname == "precise" else 0.25) * info["iterations"]
    if mname == "precise" and sname == "redblack":
        assert np.array_equal(got, want) and its == info["iterations"] and delta == info["delta"], \
            "the default configuration must BE the reference's iteration"


def test_8192_squared_tol_jacobi_against_the_reference_identical_mode(record_property):
    """BASELINE configs[2] at full size, both relaxed to eps = 1e-6 on the device."""
    m = [8192, 8192]
    u0, locked = synthetic_grid(m)
    ref, rits, rdelta = relax(m, u0, locked, eh.MATH_PRECISE, eh.SCHEME_REDBLACK)
    ref = ref.copy()
    got, its, delta = relax(m, u0, locked, eh.MATH_TOL, eh.SCHEME_JACOBI)
    rel, ab = distance(got, ref, locked)
    record_property("max_rel", rel)
    record_property("max_abs", ab)
    print("8192^2: precise red-black %d iterations (delta %.3e), tol Jacobi %d (delta %.3e); max rel %.3e, max abs %.3e, u in [%.1f, %.2f]"
          % (rits, rdelta, its, delta, rel, ab, float(ref[ref > -9e5].min()), float(ref[(ref > -9e5) & (locked == 0)].max())))
    assert rdelta < 1e-6 and delta < 1e-6
    assert rel <= BAR
    assert abs(its - rits) <= 0.25 * rits


def test_512_cubed_default_relaxation_equals_the_cpu_statement_of_the_reference(record_property):
    """BASELINE configs[4] at full size: the library's defaults (precise arithmetic, the reference's red-black half-sweeps of the 7-point stencil)
    relaxed to eps = 1e-6 on the device against the reference's loop run on the CPU at this size (round 6: tests/golden/generate_8192_golden.py
    --cube 512 -- harmonic_complete_cpu's loop with the half-sweeps dealt to threads, the sequential result bit for bit, pinned against the reference's
    own 3-D goldens by tests/test_oracle.py): iteration count, delta, 16 384 samples and the sha256 of all 134 217 728 cells.  And the timed tol
    arithmetic's converged field against that same field."""
    import hashlib
    import json

    golden = os.path.join(O.ROOT, "tests", "golden", "synthetic_512cubed.json")
    if not os.path.exists(golden):
        pytest.skip("tests/golden/synthetic_512cubed.json not generated")
    g = json.load(open(golden))
    m = [512, 512, 512]
    u0, locked = synthetic_grid(m)
    assert g["sha_u0"] == hashlib.sha256(u0.tobytes()).hexdigest() and g["sha_locked"] == hashlib.sha256(locked.tobytes()).hexdigest()
    ref, rits, rdelta = relax(m, u0, locked, eh.MATH_PRECISE, eh.SCHEME_REDBLACK)
    ref = ref.copy()
    assert (rits, rdelta) == (g["iterations"], g["delta"])
    assert np.array_equal(ref[np.asarray(g["sample_index"])], np.asarray(g["sample_u"], dtype=np.float32))
    assert hashlib.sha256(ref.tobytes()).hexdigest() == g["sha_u"], "the device's converged 512^3 field differs from the CPU statement of the reference's"
    got, its, delta = relax(m, u0, locked, eh.MATH_TOL, eh.SCHEME_JACOBI)
    rel, ab = distance(got, ref, locked)
    record_property("max_rel", rel)
    assert delta < 1e-6 and rel <= BAR


@pytest.mark.parametrize("name", ["g2d_64", "g2d_70x66_dense", "g2d_8x300", "g3d_16", "g3d_20x12x34"])
def test_empty_environment_runs_the_reference_iteration(goldens, name):
    """No EPIC_HIP_* variable at all (what the ROS plugin's process looks like): harmonic_complete_gpu must produce the
    reference's converged field bit for bit, after the reference's number of iterations, with its final delta."""
    g, info = goldens["small"], goldens["manifest"]["small"][name]
    with scheme_env(None):
        assert "EPIC_HIP_MATH" not in os.environ and "EPIC_HIP_SCHEME" not in os.environ
        h = Harmonic()
        h.set_grid(g[name + "/m"], g[name + "/u0"], g[name + "/locked"])
        h.epsilon = info["epsilon"]
        h.numIterationsToStaggerCheck = info["stagger"]
        assert E.harmonic_complete_gpu(h, 1024) == 0
    assert h.currentIteration == info["iterations"] and float(h.delta) == info["delta"]
    assert np.array_equal(h.u_array().ravel(), np.ravel(g[name + "/converged"]))
