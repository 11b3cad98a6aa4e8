cd $GRAFT_REPO_ROOT
for m in df32 fast traffic; do echo "== jacobi $m"; python bench.py --no-cpu --no-relax --math $m 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["launch_us"], d["value"])'; done
echo "== redblack precise"; python bench.py --no-cpu --no-relax --scheme redblack 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["launch_us"], d["value"], d["ms_per_step"])'
echo "== redblack df32"; python bench.py --no-cpu --no-relax --scheme redblack --math df32 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["launch_us"], d["value"], d["ms_per_step"])'
echo "== 512^3 jacobi"; python tools/bench_config.py --grid 512 512 512 --sweeps 200 --relax 2>/dev/null | tail -1
echo "== 512^3 redblack"; EPIC_HIP_SCHEME=redblack python tools/bench_config.py --grid 512 512 512 --sweeps 200 --relax 2>/dev/null | tail -1
echo "== 32768^2 jacobi"; python tools/bench_config.py --grid 32768 32768 --sweeps 30 2>/dev/null | tail -1
