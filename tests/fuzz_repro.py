"""Re-run ONE case of a campaign of tests/fuzz_gpu_parity.py (the draws are sequential: the cases before it are drawn and skipped), with
optional overrides of its environment, and say where the library's read-backs leave the checker's.

    python tests/fuzz_repro.py --node --seed 61 --case 424 [--set EPIC_HIP_DEFER=0 --unset EPIC_HIP_TRACK] [--cut N]
--cut N: only the first N operations of the script (followed by a read-back)."""
import argparse
import sys

import numpy as np

import fuzz_gpu_parity as F


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, required=True)
    ap.add_argument("--case", type=int, required=True)
    ap.add_argument("--node", action="store_true")
    ap.add_argument("--set", action="append", default=[])
    ap.add_argument("--unset", action="append", default=[])
    ap.add_argument("--cut", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    for i in range(a.case + 1):
        case = F.node_case(rng) if a.node else F.draw_case(rng)
    if not a.node:
        sys.exit("only --node cases so far")
    m, u0, locked, mode, env, ops = case
    for kv in a.set:
        k, v = kv.split("=", 1)
        env[k] = v
    for k in a.unset:
        env[k] = None
    if a.cut:
        ops = list(ops[:a.cut]) + ["r"]
    print("grid", m, "mode", mode[0], "env", {k: v for k, v in env.items() if v is not None})
    print("script", F.script_text(ops))
    want = F.checker_script(m, u0, locked, mode, ops)
    got = F.library_script(m, u0, locked, mode, env, ops)
    for j, (g, w) in enumerate(zip(got, want)):
        diff = np.flatnonzero(g[0] != w[0])
        print("read-back %d: %d cells differ, delta library %r checker %r%s" % (j, diff.size, g[1], w[1], "" if not diff.size else
              "; first at %s: %r vs %r; max |d| %.3e" % (np.unravel_index(diff[0], m), g[0][diff[0]], w[0][diff[0]], np.abs(g[0][diff] - w[0][diff]).max())))
    print("path:", (F.LAST["dump"] or {}).get("path", {}).get("plain_batch"))


if __name__ == "__main__":
    main()
