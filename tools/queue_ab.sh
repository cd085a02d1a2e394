#!/bin/bash
# Run ON THE GPU BOX: the chunk-queue form of large launches against the one-stream-per-wave form (AECM_QUEUE_CHUNK = chunk
# length in blocks, 0 = off), interleaved (profiles/r04_experiments.md section 1).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
run() {
  q=$1; shift
  AECM_QUEUE_CHUNK=$q timeout 300 python bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-8} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk $q $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step')"
}
for rep in 1 2; do
for q in ${CHUNKS:-0 32 64 128 256}; do run $q --streams 65536 --blocks 1280; done
for q in ${CHUNKS:-0 32 64 128 256}; do run $q --streams 16384 --blocks 1280; done
for q in 0 128; do run $q --streams 32768 --blocks 1280 --fs 8000; done
for q in 0 128; do run $q --streams 65536 --blocks 1280 --clean; done
done
