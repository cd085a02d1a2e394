#!/bin/bash
# Run ON THE GPU BOX: same data (T = 1280 blocks per step), launch length varied (profiles/r04_experiments.md section 1).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
run() {
  timeout 300 python bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-8} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step')"
}
for rep in 1 2; do
run --streams 65536 --blocks 1280
run --streams 65536 --blocks 1280 --launch-blocks 640
run --streams 65536 --blocks 1280 --launch-blocks 320
run --streams 65536 --blocks 1280 --launch-blocks 160
run --streams 65536 --blocks 1280 --launch-blocks 40
run --streams 65536 --blocks 2560
run --streams 16384 --blocks 1280
run --streams 16384 --blocks 1280 --launch-blocks 160
done
python bench.py --no-cpu-baseline --steps 3 --warmup 1 --launch-blocks 160 | tail -1 | cut -c1-300
