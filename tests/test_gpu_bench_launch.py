"""`python3 bench.py --gpus N` typed as is -- no torch.distributed.run in front, the way the driver's scaling run types it --
must start its own ranks and print ONE line.  On the one GPU of this box the ranks share the device and trade their halo rows
over gloo (EPIC_BENCH_BACKEND=gloo), and the in-library leg runs its slabs on device 0 twice (EPIC_BENCH_DEVLIST=0,0): the same
code paths as on a node with N devices, minus the links."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


def test_plain_command_with_two_gpus_launches_its_own_ranks():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "EPIC_HIP_SCHEME", "EPIC_HIP_MATH")}
    env.update({"EPIC_BENCH_BACKEND": "gloo", "EPIC_BENCH_DEVLIST": "0,0"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--develop", "2000"], env=env, capture_output=True, text=True, timeout=840, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]       # ONE JSON line, nothing else on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["value"] > 0 and d["ms_per_step"] > 0
    ranks = d["ranks"]
    assert ranks["ranks_seen"] == 2 and ranks["backend"] == "gloo"
    assert sorted(x["rank"] for x in ranks["per_rank"]) == [0, 1] and len({x["pid"] for x in ranks["per_rank"]}) == 2
    assert "in_library" in d and "error" not in d["in_library"], d.get("in_library")
    assert d["in_library"]["slabs"] == 2 and d["in_library"]["value"] > 0
    assert d["roofline"]["bound"] in ("hbm", "valu") and d["roofline"]["frac"] > 0
