#!/bin/bash
# Run ON THE GPU BOX: the pipelined form of launches the chip holds at once (AECM_PIPELINED: 0 = off, n = from n streams)
# against the one-stream-per-wave kernels, interleaved (profiles/r04_experiments.md section 3).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
run() {
  q=$1; shift
  AECM_PIPELINED=$q timeout 300 python bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelined $q $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step', d['roofline']['kernel'])"
}
for rep in 1 2; do
for s in ${SIZES:-4096 3072 2048 1024 256 64}; do
  for q in 0 2; do run $q --streams $s --blocks 2048; done
done
for q in 0 2; do run $q --streams 4096 --blocks 2048 --fs 8000; done
done
