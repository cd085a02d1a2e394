#!/bin/bash
# Run ON THE GPU BOX: bench.py (kernel only) for A/B library variants, REPS interleaved repetitions.
#   tools/ab_bench.sh variant1 variant2 ...     (webrtc_aecm_amd/_lib/ab_<v>.so; BENCH_ARGS / REPS from the environment)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    AECM_LIB_PATH=webrtc_aecm_amd/_lib/ab_$v.so timeout 180 python bench.py --no-cpu-baseline --steps 10 --warmup 2 ${BENCH_ARGS:-} 2>&1 | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms')"
  done
done
