"""The reference's own ctypes wrapper, unchanged, on this repository's libepic.so.

tests/golden/ref_wrapper_binding.json is what tests/golden/generate_ref_wrapper_binding.py recorded in the build container: the
reference's libepic/python/epic/epic_harmonic.py imported as it is with its one CDLL call answered by epic_amd/lib/libepic.so --
all thirty `argtypes` lines bound --, and the reference's Harmonic.solve() run through it.  Here (no /root/reference needed) the
fixture is held against the library as built now and against the goldens the reference's C sources produced; where the reference
is present the script is run again and must reproduce the fixture."""
import ctypes as ct
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "ref_wrapper_binding.json")


@pytest.fixture(scope="module")
def binding():
    return json.load(open(FIXTURE))


def test_every_symbol_the_reference_wrapper_binds_is_exported(binding):
    lib = ct.CDLL(os.path.join(ROOT, "epic_amd", "lib", "libepic.so"))
    assert binding["n_symbols"] == 30 == len(binding["symbols_bound"])
    for name in binding["symbols_bound"]:
        assert hasattr(lib, name), name
    assert binding["wrapper"]["answered_with"] == "epic_amd/lib/libepic.so"
    assert binding["wrapper"]["asked_for"].endswith("lib/libepic.so")


def test_the_reference_wrapper_solved_the_goldens_through_this_library(binding):
    g = np.load(os.path.join(ROOT, "tests", "golden", "small_grids.npz"))
    manifest = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["small"]
    assert sorted(binding["results"]) == ["g2d_32 cpu", "g2d_32 gpu", "g3d_8 cpu", "g3d_8 gpu"]
    for key, r in binding["results"].items():
        name = key.split()[0]
        want = np.asarray(g[name + "/converged"], dtype=np.float32)
        assert r["equals_reference_golden"] is True and r["sha256_u"] == hashlib.sha256(want.tobytes()).hexdigest(), key
        assert r["iterations"] == manifest[name]["iterations"] and r["delta"] == manifest[name]["delta"], key
        assert r["device_pointers_null_afterwards"] is True, key


@pytest.mark.skipif(not os.path.isdir("/root/reference/libepic/python/epic"), reason="the reference is only in the build container")
def test_the_script_reproduces_the_fixture_here(binding, tmp_path):
    before = open(FIXTURE).read()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "generate_ref_wrapper_binding.py")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    after = json.load(open(FIXTURE))
    try:
        assert after["symbols_bound"] == binding["symbols_bound"] and after["results"] == binding["results"]
        assert after["wrapper"]["sha256_epic_harmonic"] == binding["wrapper"]["sha256_epic_harmonic"]
    finally:
        open(FIXTURE, "w").write(before)    # (gpu_visible may differ between hosts: keep the committed text)
