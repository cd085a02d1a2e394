#!/bin/bash
# Run ON THE GPU BOX: bench.py (kernel only) on the libraries named on the command line (paths), REPS interleaved repetitions.
#   tools/ab_quick.sh webrtc_aecm_amd/_lib/libaecm_mi355x.so gpurun_out/base.so ...      (BENCH_ARGS / REPS / STEPS from the environment)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
for rep in $(seq 1 ${REPS:-3}); do
  for lib in "$@"; do
    AECM_LIB_PATH=$lib timeout 180 python bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-8} --warmup 2 ${BENCH_ARGS:-} 2>&1 | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib)', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms')"
  done
done
