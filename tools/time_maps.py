"""Wall time of the complete relaxation of the reference's maps (BASELINE configs 0-1 and the other fixtures) through
harmonic_complete_gpu, as the ROS plugin calls it (src/epic_nav_core_plugin.cpp:256): upload, relaxation, download.

    python tools/time_maps.py [--maps basic,maze,umass] [--eps 1e-6] [--modes default,tol_rb,tol_jacobi,jacobi]
                              [--tile 1,0] [--halo 8,12,16] [--rows 0]
Prints one line per configuration and a JSON summary on the last line.
"""
import argparse
import json
import os

os.environ.setdefault("EPIC_HIP_STUDY", "1")   # this tool steers the kernel plan with study knobs (epic_amd/csrc/driver_config.cpp)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epic_amd.harmonic_map import HarmonicMap  # noqa: E402

MODES = {  # name -> (EPIC_HIP_MATH, EPIC_HIP_SCHEME); None = variable absent (the library's default)
    "default": (None, None),
    "jacobi": (None, "jacobi"),
    "tol_rb": ("tol", "redblack"),
    "tol_jacobi": ("tol", "jacobi"),
}


def setenv(k, v):
    if v is None:
        os.environ.pop(k, None)
    else:
        os.environ[k] = str(v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maps", default="basic,maze,umass")
    ap.add_argument("--eps", default="1e-6")
    ap.add_argument("--modes", default="default,tol_rb,tol_jacobi")
    ap.add_argument("--tile", default="1,0")
    ap.add_argument("--halo", default="auto", help="ghost rings per tile launch: numbers, or auto = the library's choice")
    ap.add_argument("--rows", default="0")
    ap.add_argument("--repeat", type=int, default=2)
    args = ap.parse_args()
    ref = {}
    for f in ("manifest.json", "ref_maps.json"):
        p = os.path.join(ROOT, "tests/golden", f)
        if os.path.exists(p):
            for k, v in json.load(open(p))["maps"].items():
                ref.setdefault(k, {}).update(v.get("runs", {}))
    out = []
    for eps in [float(x) for x in args.eps.split(",")]:
        for mode in args.modes.split(","):
            setenv("EPIC_HIP_MATH", MODES[mode][0])
            setenv("EPIC_HIP_SCHEME", MODES[mode][1])
            for name in args.maps.split(","):
                for tile in args.tile.split(","):
                    for halo in (args.halo.split(",") if tile == "1" else ["0"]):
                        for rows in (args.rows.split(",") if tile == "1" else ["0"]):
                            setenv("EPIC_HIP_TILE", tile)
                            setenv("EPIC_HIP_TILE_HALO", halo if tile == "1" and halo != "auto" else None)
                            setenv("EPIC_HIP_TILE_ROWS", rows if rows != "0" else None)
                            best = None
                            for _ in range(1 + args.repeat):   # the first run warms up (code load, graph capture)
                                h = HarmonicMap().load(os.path.join(ROOT, "tests/golden/maps", name + ".png"))
                                t0 = time.time()
                                h.solve(process="gpu", epsilon=eps)
                                wall = time.time() - t0   # initialise x3 + complete + uninitialise, as a caller sees it
                                best = wall if best is None or _ == 1 else min(best, wall)
                            r = ref.get(name, {}).get(f"{eps:g}", {})
                            rec = dict(map=name, shape=list(h.shape), eps=eps, mode=mode, tile=int(tile), halo=halo, tile_rows=int(rows),
                                       iterations=int(h.currentIteration), seconds=round(best, 4),
                                       us_per_iteration=round(best / h.currentIteration * 1e6, 3),
                                       reference_cpu_seconds=r.get("seconds"), reference_iterations=r.get("iterations"))
                            out.append(rec)
                            print(f"{mode:10s} eps {eps:g} {name:14s} {str(list(h.shape)):12s} tile {tile} halo {halo:>4s} rows {rows:>2s}: "
                                  f"{rec['iterations']:7d} iterations {best:.4f} s ({rec['us_per_iteration']:.2f} us/iteration)"
                                  f"  reference CPU {r.get('seconds')} s ({r.get('iterations')})", flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
