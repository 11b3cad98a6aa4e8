"""Wall time of the complete relaxation of the reference maps (BASELINE configs 1-2) through harmonic_complete_gpu."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import json
from epic_amd.harmonic_map import HarmonicMap
g = json.load(open(os.path.join(ROOT, "tests/golden/manifest.json")))["maps"]
for scheme in ("jacobi", "redblack"):
    os.environ["EPIC_HIP_SCHEME"] = scheme
    for name in ("basic", "maze", "umass"):
        h = HarmonicMap().load(os.path.join(ROOT, "tests/golden/maps", name + ".png"))
        h.solve(process="gpu", epsilon=1e-6)       # warm-up (graph capture, code load)
        h = HarmonicMap().load(os.path.join(ROOT, "tests/golden/maps", name + ".png"))
        wall, _ = h.solve(process="gpu", epsilon=1e-6)
        ref = g[name]["runs"]["1e-06"]
        print(f"{scheme:9s} {name:6s} {list(h.shape)} iterations {h.currentIteration:6d} gpu {wall:.3f} s "
              f"({wall / h.currentIteration * 1e6:.2f} us/iteration)  reference CPU {ref['seconds']} s ({ref['iterations']} half-sweeps)", flush=True)
