#!/bin/bash
# Run ON THE GPU BOX: how much of a launch is its drain?  Frame rate against streams per launch (S) and blocks per
# launch (T) with S x T fixed or not (profiles/r04_experiments.md section 1).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
run() {
  timeout 300 python bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-8} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step')"
}
for rep in 1 2; do
run --streams 65536 --blocks 1280
run --streams 131072 --blocks 640
run --streams 262144 --blocks 320
run --streams 65536 --blocks 320
run --streams 65536 --blocks 160
run --streams 16384 --blocks 1280
run --streams 14336 --blocks 1280
run --streams 7168 --blocks 1280
run --streams 28672 --blocks 1280
run --streams 32768 --blocks 1280
done
