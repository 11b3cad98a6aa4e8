"""The committed record of the tol mode's parity campaign (tests/golden/tol_campaign.json, written by tests/tol_campaign.py): the
benchmarked arithmetic's "until converged" loop -- tol iterations, then the reference's own from the hand-over on -- against the
reference's harmonic_complete_cpu on generated maps the hand-over rule was never tuned on.

No GPU: the loop compared is the checker's statement (oracle/tol_checker.c: oracle_tol_complete), which tests/test_gpu_tol.py and the
fuzz campaigns hold the device's loop to bit for bit.  Here: the record is what the generator and the checker produce TODAY (a sample is
re-run and compared number for number), its summary follows from its cases, and it says what DESIGN.md section 2 quotes."""
import json
import os

import numpy as np
import pytest

import tol_campaign as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORD = os.path.join(ROOT, "tests", "golden", "tol_campaign.json")


@pytest.fixture(scope="module")
def record():
    return json.load(open(RECORD))


def test_the_summary_follows_from_the_cases(record):
    cases = [dict(c) for c in record["cases"]]
    s = T.summarise(cases)
    assert json.loads(json.dumps(s)) == record["summary"]
    assert s["cases"] == len(cases) >= 300 and s["maps"] * len(T.EPSILONS) * len(T.SCHEMES) == s["cases"]
    assert set(c["family"] for c in cases) == set(T.FAMILIES)
    assert all(c["rc"] == [0, 0] for c in cases)                      # every run converged, on both sides
    assert all(c["unreached_and_locked_exact"] for c in cases)        # cells the front never reaches and locked cells: exact


def test_what_the_record_says(record):
    """The statements DESIGN.md section 2 and the bench line quote: no miss with the library's red-black scheme; every miss is a Jacobi
    run stopped at the reference's own iteration count at epsilon = 1e-2 (second-chain lag, tests/tol_campaign.py: explain); nothing
    unexplained; relaxations to stagnation (1e-6) all inside the bar by a factor of five."""
    s = record["summary"]
    assert s["misses_redblack"] == 0 and s["misses_unexplained"] == 0
    assert s["misses"] == s["misses_by_explanation"].get("jacobi_second_chain", 0) <= 0.02 * s["cases"]
    assert s["by_epsilon"]["1e-06"]["misses"] == 0 and s["by_epsilon"]["1e-06"]["worst_rel"] < 2e-6
    assert s["by_epsilon"]["0.001"]["misses"] == 0
    for c in record["cases"]:
        if not c["within_bar"]:
            assert c["scheme"] == "jacobi" and c["epsilon"] == 1e-2 and c["iterations"] == c["reference_iterations"] and c["max_rel"] < 1e-3
    # at the callers' epsilons the loop stops where the reference stops
    for eps in ("0.01", "0.001"):
        assert s["by_epsilon"][eps]["same_iterations"] == s["by_epsilon"][eps]["cases"]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("family,seed", [("dense", 1407), ("labyrinth", 1600), ("maze", 1103), ("cube", 1708)])
def test_a_sample_re_run_today_gives_the_recorded_numbers(record, family, seed):
    """Generator and checker are pinned by the record: the same map (sha of its mask), the same iteration counts, hand-over iteration and
    distance, for all six (epsilon, scheme) runs of the map -- two of them recorded misses (dense 1407 and the 3-D cube 1708, Jacobi, 1e-2)."""
    want = [c for c in record["cases"] if c["family"] == family and c["seed"] == seed]
    assert len(want) == 6
    got = T.run_map((family, seed))
    key = lambda c: (-c["epsilon"], c["scheme"])
    for g, w in zip(sorted(got, key=key), sorted(want, key=key)):
        for k in ("m", "free_cells", "reached_cells", "epsilon", "scheme", "reference_iterations", "iterations", "finish_from", "plateau_warning",
                  "identical", "within_bar"):
            assert g[k] == w[k], (k, g[k], w[k])
        assert g["max_rel"] == pytest.approx(w["max_rel"], rel=1e-12, abs=0.0)


REFCHECKS = os.path.join(ROOT, "tests", "golden", "tol_campaign_reference_checks.json")


def test_the_jacobi_scheme_with_reference_checks_misses_nothing():
    """The same 120 maps x 3 epsilons for the Jacobi scheme with EPIC_HIP_JACOBI_CHECKS=reference (tests/tol_campaign.py --ref-checks; every check
    iteration the reference's half-sweep: tests/test_jacobi_reference_checks.py): the second chain's lag is gone -- no case outside the bar, the worst a
    fifth of the plain Jacobi checks' worst INSIDE it, and at the callers' epsilons every run stops at the reference's iteration."""
    rec = json.load(open(REFCHECKS))
    cases = [dict(c) for c in rec["cases"]]
    s = T.summarise(cases)
    assert json.loads(json.dumps(s)) == rec["summary"]
    assert s["cases"] == 360 and s["maps"] == 120 and all(c["scheme"] == "jacobi+reference_checks" and c["rc"] == [0, 0] for c in cases)
    assert s["misses"] == 0 and s["worst_rel"] < 2e-6 and s["warnings"] == 0
    assert s["by_epsilon"]["0.01"]["worst_rel"] < 1e-6 and s["by_epsilon"]["0.001"]["worst_rel"] < 1e-6
    for eps in ("0.01", "0.001"):
        assert s["by_epsilon"][eps]["same_iterations"] == 120
    main = {(c["family"], c["seed"], c["epsilon"]): c for c in json.load(open(RECORD))["cases"] if c["scheme"] == "jacobi"}
    assert all(main[(c["family"], c["seed"], c["epsilon"])]["reference_iterations"] == c["reference_iterations"] for c in cases)   # the same maps, the same reference runs


@pytest.mark.timeout(600)
def test_a_reference_checks_sample_re_run_today_gives_the_recorded_numbers(monkeypatch):
    rec = json.load(open(REFCHECKS))
    monkeypatch.setenv("EPIC_CAMPAIGN_REF_CHECKS", "1")
    try:
        got = T.run_map(("dense", 1407))
    finally:
        import _oracle as O

        O.oracle().oracle_set_jacobi_ref_checks(0)
    want = [c for c in rec["cases"] if c["family"] == "dense" and c["seed"] == 1407]
    assert len(got) == len(want) == 3
    for g, w in zip(sorted(got, key=lambda c: -c["epsilon"]), sorted(want, key=lambda c: -c["epsilon"])):
        for k in ("m", "epsilon", "scheme", "reference_iterations", "iterations", "finish_from", "identical", "within_bar"):
            assert g[k] == w[k], (k, g[k], w[k])
        assert g["max_rel"] == pytest.approx(w["max_rel"], rel=1e-12, abs=0.0)


def test_the_generated_maps_are_the_same_everywhere():
    """numpy's Generator streams are stable across versions for the calls used; a changed map would silently change the campaign."""
    import hashlib

    sums = {}
    for f in T.FAMILIES:
        m, u0, locked = T.make_case(f, 1000 + 100 * T.FAMILIES.index(f))
        sums[f] = (m, hashlib.sha256(locked.tobytes()).hexdigest()[:16], int((u0 == 0.0).sum()))
    assert sums == EXPECTED_MAPS, sums


EXPECTED_MAPS = {'corridor': ([156, 135], '8b09be044d741990', 1),
 'cube': ([16, 20, 64], '1b857df62d86812d', 3),
 'dense': ([62, 102], 'dc3779361d686ce5', 4),
 'labyrinth': ([43, 109], '0262ee41a91d1218', 1),
 'maze': ([118, 77], '4137f1bdd2a2e35e', 3),
 'office': ([82, 67], 'b2637886bc1fe4ef', 4),
 'rooms': ([116, 223], 'e92183ea04321686', 5),
 'sparse': ([219, 357], '867e42d825e10156', 2)}
