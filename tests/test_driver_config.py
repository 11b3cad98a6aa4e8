"""epic_amd/csrc/driver_config.cpp -- the one place where the library reads its environment -- on the CPU: a three-line harness is
compiled against it with g++ and prints Config::from_env().json() for a few environments."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "epic_amd", "csrc")
HARNESS = r'''
#include <stdio.h>
#include "driver_config.h"
int main() {
    printf("%s\n", epic_drv::Config::from_env().json().c_str());
    const epic_hip::LaunchKnobs &k = epic_hip::process_launch_knobs();
    printf("{\"flags\": %d, \"list_waves\": %zu, \"pair3d\": %d, \"pair3d_rows\": %d, \"march_x0\": %d}\n", k.flags, k.list_waves, (int)k.pair3d, k.pair3d_rows, (int)k.march_x0);
    return 0;
}
'''

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("cfg")
    src = d / "main.cpp"
    src.write_text(HARNESS)
    exe = str(d / "cfg")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", CSRC, str(src), os.path.join(CSRC, "driver_config.cpp"), "-lpthread", "-o", exe], check=True)
    return exe


def run(exe, **env):
    e = {k: v for k, v in os.environ.items() if not k.startswith("EPIC_HIP_")}
    e.update({k: str(v) for k, v in env.items()})
    out = subprocess.run([exe], env=e, capture_output=True, text=True, check=True).stdout.splitlines()
    return json.loads(out[0]), json.loads(out[1])


def test_defaults_are_the_librarys(harness):
    c, k = run(harness)
    assert c["study"] is False and c["jacobi_checks"] == "jacobi"
    assert c["math"] == 0 and c["scheme"] == "redblack" and c["track_mode"] == 2 and c["rows_per_task"] == 0      # the reference's iteration, bit-exact
    assert c["devices"] == "" and c["halo"] == 0 and c["threads"] is True and c["spin_us"] == 20 and c["no_peer"] is False
    assert c["no_fuse"] is False and c["no_graph"] is False and c["fuse_min_cells"] == -1 and c["tune"] is True      # -1: by arithmetic and scheme (driver_plan.hip)
    assert c["tile"] is True and c["tile_max_cells"] == -1 and c["tile_pipeline"] is True and c["track_pairs"] is True and c["defer"] is True
    assert c["track_switch"] == -1 and c["tol_finish"] == -1 and c["tol_finish_factor"] == 0
    assert k == {"flags": 7, "list_waves": 0, "pair3d": 1, "pair3d_rows": 0, "march_x0": 0}


def test_every_knob_is_parsed(harness):
    c, k = run(harness, EPIC_HIP_STUDY="1", EPIC_HIP_MATH="tol", EPIC_HIP_SCHEME="jacobi", EPIC_HIP_JACOBI_CHECKS="reference", EPIC_HIP_TRACK="1", EPIC_HIP_ROWS_PER_TASK="12", EPIC_HIP_DEVICES="0,1,1,3",
               EPIC_HIP_HALO="5", EPIC_HIP_NO_PEER="1", EPIC_HIP_THREADS="0", EPIC_HIP_SPIN_US="0", EPIC_HIP_NO_FUSE="1", EPIC_HIP_NO_GRAPH="1",
               EPIC_HIP_FUSE_MIN_CELLS="0", EPIC_HIP_FUSED_ROWS="33", EPIC_HIP_TUNE="0", EPIC_HIP_TILE="0", EPIC_HIP_TILE_MAX_CELLS="1000",
               EPIC_HIP_TILE_ROWS="7", EPIC_HIP_TILE_WIDTH="128", EPIC_HIP_TILE_HALO="9", EPIC_HIP_TILE_PIPELINE="0", EPIC_HIP_DEFER="0", EPIC_HIP_TRACK_PAIRS="0",
               EPIC_HIP_TRACK_PAIR_ROWS="6", EPIC_HIP_TRACK_SWITCH="0.5", EPIC_HIP_TOL_FINISH="0", EPIC_HIP_TOL_FINISH_FACTOR="30",
               EPIC_HIP_FLAGS="2", EPIC_HIP_LIST_WAVES="512", EPIC_HIP_3D_PAIR="0", EPIC_HIP_3D_PAIR_ROWS="20", EPIC_HIP_3D_MARCH="x0")
    assert c["math"] == 4 and c["scheme"] == "jacobi" and c["track_mode"] == 1 and c["rows_per_task"] == 12 and c["devices"] == "0,1,1,3"
    assert c["jacobi_checks"] == "reference"
    assert c["halo"] == 5 and c["no_peer"] is True and c["threads"] is False and c["spin_us"] == 0 and c["no_fuse"] is True and c["no_graph"] is True
    assert c["fuse_min_cells"] == 0 and c["fused_rows"] == 33 and c["tune"] is False and c["tile"] is False and c["tile_max_cells"] == 1000
    assert (c["tile_rows"], c["tile_width"], c["tile_halo"]) == (7, 128, 9) and c["tile_pipeline"] is False and c["track_pairs"] is False
    assert c["track_pair_rows"] == 6 and c["track_switch"] == 0.5 and c["tol_finish"] == 0 and c["tol_finish_factor"] == 30 and c["defer"] is False
    assert k == {"flags": 2, "list_waves": 512, "pair3d": 0, "pair3d_rows": 20, "march_x0": 1}
    assert (c["flags"], c["list_waves"], c["pair3d"], c["pair3d_rows"], c["march_x0"]) == (2, 512, False, 20, True)


def test_values_out_of_range_fall_back(harness):
    c, _ = run(harness, EPIC_HIP_STUDY="1", EPIC_HIP_MATH="double", EPIC_HIP_SCHEME="sor", EPIC_HIP_TRACK="7", EPIC_HIP_HALO="0", EPIC_HIP_SPIN_US="-3",
               EPIC_HIP_TOL_FINISH_FACTOR="0.5", EPIC_HIP_LIST_WAVES="2", EPIC_HIP_TILE_HALO="-1", EPIC_HIP_FUSED_ROWS="0")
    assert c["math"] == 0 and c["scheme"] == "redblack" and c["track_mode"] == 2 and c["halo"] == 0 and c["spin_us"] == 20
    assert c["tol_finish_factor"] == 0 and c["list_waves"] == 0 and c["tile_halo"] == 0 and c["fused_rows"] == 0
    c, _ = run(harness, EPIC_HIP_DEVICES="0,x")       # a malformed list is kept as text (the registry warns and ignores it) ...
    assert c["devices"] == "0,x"
    c, _ = run(harness, EPIC_HIP_SPIN_US="99999999")   # ... and a spin is never longer than 0.1 s
    assert c["spin_us"] == 100000


def test_whatever_the_environment_holds_the_dump_stays_one_json_object(harness):
    """EPIC_HIP_DEVICES is the caller's text and is echoed in the dump: quotes and backslashes are escaped, control characters blanked, a
    very long value clipped (round 6; until then a fixed buffer truncated the object and a quote broke it -- bench.py parses this)."""
    nasty = '0,"1\\' + "\t" + "x" * 5000
    c, _ = run(harness, EPIC_HIP_DEVICES=nasty)
    assert c["devices"].startswith('0,"1\\ x') and len(c["devices"]) == 256
    assert c["math"] == 0 and c["march_x0"] is False       # the fields behind the text are all there


def test_study_knobs_are_read_only_when_asked_for(harness):
    """Two classes of variables (round 6, driver_config.cpp).  PRODUCT knobs -- arithmetic, scheme, work lists, devices and their transport, the
    tol mode's finishing iterations, deferred updates -- are always honoured.  STUDY knobs -- thresholds, task heights, tile plans, launch flags,
    the tuner: they select a code path, never a result -- are what tests, fuzz campaigns and bench.py's A/B legs steer the library with; a drop-in
    library in somebody else's process reads them only under EPIC_HIP_STUDY=1, and says once on stderr that it ignored one otherwise."""
    study = dict(EPIC_HIP_ROWS_PER_TASK="12", EPIC_HIP_NO_FUSE="1", EPIC_HIP_NO_GRAPH="1", EPIC_HIP_FUSE_MIN_CELLS="0", EPIC_HIP_FUSED_ROWS="33",
                 EPIC_HIP_TUNE="0", EPIC_HIP_TILE="0", EPIC_HIP_TILE_MAX_CELLS="1000", EPIC_HIP_TILE_ROWS="7", EPIC_HIP_TILE_WIDTH="128",
                 EPIC_HIP_TILE_HALO="9", EPIC_HIP_TILE_PIPELINE="0", EPIC_HIP_TRACK_PAIRS="0", EPIC_HIP_TRACK_PAIR_ROWS="6", EPIC_HIP_TRACK_SWITCH="0.5",
                 EPIC_HIP_TOL_FINISH_FACTOR="30", EPIC_HIP_FLAGS="2", EPIC_HIP_LIST_WAVES="512", EPIC_HIP_3D_PAIR="0", EPIC_HIP_3D_PAIR_ROWS="20",
                 EPIC_HIP_3D_MARCH="x0")
    product = dict(EPIC_HIP_MATH="tol", EPIC_HIP_SCHEME="jacobi", EPIC_HIP_JACOBI_CHECKS="reference", EPIC_HIP_TRACK="1", EPIC_HIP_DEVICES="0,1", EPIC_HIP_HALO="5", EPIC_HIP_NO_PEER="1",
                   EPIC_HIP_THREADS="0", EPIC_HIP_SPIN_US="7", EPIC_HIP_TOL_FINISH="0", EPIC_HIP_DEFER="0")
    defaults, k0 = run(harness)
    e = {k: v for k, v in os.environ.items() if not k.startswith("EPIC_HIP_")}
    e.update(study)
    e.update(product)
    r = subprocess.run([harness], env=e, capture_output=True, text=True, check=True)
    c, k = json.loads(r.stdout.splitlines()[0]), json.loads(r.stdout.splitlines()[1])
    assert c["study"] is False and k == k0
    for key in ("rows_per_task", "no_fuse", "no_graph", "fuse_min_cells", "fused_rows", "tune", "tile", "tile_max_cells", "tile_rows", "tile_width",
                "tile_halo", "tile_pipeline", "track_pairs", "track_pair_rows", "track_switch", "tol_finish_factor", "flags", "list_waves", "pair3d",
                "pair3d_rows", "march_x0"):
        assert c[key] == defaults[key], key
    assert (c["math"], c["scheme"], c["track_mode"], c["devices"], c["halo"], c["no_peer"], c["threads"], c["spin_us"], c["tol_finish"], c["defer"]) == \
        (4, "jacobi", 1, "0,1", 5, True, False, 7, 0, False)
    assert c["jacobi_checks"] == "reference"
    assert r.stderr.count("is a study knob and is ignored without EPIC_HIP_STUDY=1") == 1
    c2, _ = run(harness, EPIC_HIP_STUDY="1", **study)
    assert c2["study"] is True and c2["no_fuse"] is True and c2["tile_rows"] == 7 and c2["flags"] == 2
    c3, _ = run(harness, EPIC_HIP_STUDY="0", **study)
    assert c3["study"] is False and c3["no_fuse"] is False
