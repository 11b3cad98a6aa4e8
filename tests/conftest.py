import os
import subprocess
import sys

import pytest

# The checkers' OpenMP teams (oracle/) default to one thread per hardware thread of the HOST; a GPU box gives this process
# a 16-CPU share of a much larger machine, where such a team spends its time in barriers.  Set before libgomp is loaded.
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

# The library's default iteration is the reference's red-black half-sweep (bit-identical to harmonic_complete_cpu with the
# default precise math).  Most of this suite was written against the Jacobi scheme -- the one BASELINE.json's metric names and
# bench.py times -- and compares with the checker's Jacobi, so the session selects it the way a user would; the tests of the
# DEFAULT remove the variable again (scheme_env(None) below): tests/test_gpu_bench_parity.py::test_empty_environment_...,
# tests/test_gpu_tile.py::test_maps_relax_through_tiles_..., tests/test_gpu_callers_eps.py (every reference map at the
# callers' epsilons) and the C++ plugin replay (tests/test_gpu_plugin_replay.py), which runs without any variable.
os.environ.setdefault("EPIC_HIP_SCHEME", "jacobi")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "multi_gpu: needs at least two GPUs and EPIC_TEST_MULTI_GPU=1 (always combined with gpu; skips itself otherwise)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The product library and the checker must exist before anything is imported.  Prebuilt files travel to the
    GPU box; here they are (re)built from source when missing."""
    lib = os.path.join(ROOT, "epic_amd", "lib", "libepic.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "epic_amd", "csrc")], check=True)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


class scheme_env:
    """with scheme_env("redblack"): ...   EPIC_HIP_SCHEME for the library contexts created inside, restored afterwards
    (None = variable absent: the library default)."""

    def __init__(self, scheme):
        self.scheme = scheme

    def __enter__(self):
        self.prev = os.environ.get("EPIC_HIP_SCHEME")
        if self.scheme is None:
            os.environ.pop("EPIC_HIP_SCHEME", None)
        else:
            os.environ["EPIC_HIP_SCHEME"] = self.scheme
        return self

    def __exit__(self, *exc):
        if self.prev is None:
            os.environ.pop("EPIC_HIP_SCHEME", None)
        else:
            os.environ["EPIC_HIP_SCHEME"] = self.prev
        return False


@pytest.fixture(scope="session")
def goldens():
    import json

    import numpy as np

    g = os.path.join(ROOT, "tests", "golden")
    return dict(manifest=json.load(open(os.path.join(g, "manifest.json"))),
                small=np.load(os.path.join(g, "small_grids.npz")),
                maps=np.load(os.path.join(g, "maps_converged.npz")))
