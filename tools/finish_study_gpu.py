"""Where a tol relaxation with its finishing iterations ends against the reference's own field (tests/golden/ref_maps.*), by the
factor of the hand-over rule (delta < factor * eps -> the reference's iteration): iterations, the iteration the finishing phase
started at, distance on the 16 384 samples.

    python tools/finish_study_gpu.py --maps trivial,c_space --eps 1e-6 --factors 10,100,1000 [--schemes redblack,jacobi]
"""
import argparse
import ctypes as ct
import json
import os

os.environ.setdefault("EPIC_HIP_STUDY", "1")   # this tool steers the kernel plan with study knobs (epic_amd/csrc/driver_config.cpp)
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maps", default="trivial")
    ap.add_argument("--eps", default="1e-06")
    ap.add_argument("--factors", default="10,100,1000")
    ap.add_argument("--schemes", default="redblack")
    args = ap.parse_args()
    from epic_amd import epic_harmonic as eh
    from epic_amd.harmonic_map import HarmonicMap

    E = eh._epic
    man = json.load(open(os.path.join(ROOT, "tests/golden/ref_maps.json")))
    ref = np.load(os.path.join(ROOT, "tests/golden/ref_maps.npz"))
    for name in args.maps.split(","):
        for eps in args.eps.split(","):
            run = man["maps"][name]["runs"][eps]
            idx, want = ref[name + "/sample_idx"], ref[f"{name}/samples_{eps}"]
            for scheme in args.schemes.split(","):
                for factor in args.factors.split(","):
                    os.environ["EPIC_HIP_MATH"] = "tol"
                    os.environ["EPIC_HIP_SCHEME"] = scheme
                    os.environ["EPIC_HIP_TOL_FINISH_FACTOR"] = factor
                    h = HarmonicMap().load(os.path.join(ROOT, "tests/golden/maps", name + ".png"))
                    h.epsilon = float(eps)
                    h.numIterationsToStaggerCheck = 100
                    for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
                        assert fn(h) == 0
                    t0 = time.time()
                    assert E.harmonic_execute_gpu(h, 1024) == 0
                    secs = time.time() - t0
                    fin = int(E.epic_hip_finish_iteration(h))
                    assert E.harmonic_get_potential_values_gpu(h) == 0
                    got = h.u_array().ravel()[idx]
                    d = np.abs(got.astype(np.float64) - want) / np.maximum(1.0, np.abs(want))
                    for fn in (E.harmonic_uninitialize_gpu, E.harmonic_uninitialize_dimension_size_gpu, E.harmonic_uninitialize_potential_values_gpu,
                               E.harmonic_uninitialize_locked_gpu):
                        fn(h)
                    print(f"{name} eps {eps} tol {scheme} factor {factor:>6s}: {h.currentIteration} iterations (reference {run['iterations']}), finishing from "
                          f"{fin}, max rel {d.max():.3e}, mean rel {d.mean():.3e}, {secs:.2f} s", flush=True)


if __name__ == "__main__":
    main()
