#!/bin/bash
# A/B of two builds of the library on the bench's kernels (8192^2 tol Jacobi fused pass, single sweep, precise; 512^3 tol):
#   bash tools/ab_lib.sh <libA.so> <libB.so> ...     (paths relative to the repo root; alternating rounds so that drift shows)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUNDS=${ROUNDS:-2}
for round in $(seq $ROUNDS); do
for lib in "$@"; do
  EPIC_LIB=$ROOT/$lib python3 $ROOT/bench.py --no-cpu --no-relax --no-parity --no-live-traffic --no-config4 --no-maps --steps 10 --warmup 2 2>/dev/null | python3 -c '
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=r.get("kernels",{}); g=lambda d,*ks: (g(d.get(ks[0],{}),*ks[1:]) if len(ks)>1 else d.get(ks[0])) if isinstance(d,dict) else None
print("round '$round' %-36s fused %s us (%s)  single %s  precise %s  3-D tol %s  3-D precise %s" % ("'$lib'", r["roofline"]["launch_us"], r["roofline"]["frac"], g(k,"single_sweep","launch_us"), g(k,"precise","launch_us"), g(r,"config5","us_per_sweep"), g(r,"config5","precise","us_per_sweep")))'
done
done
