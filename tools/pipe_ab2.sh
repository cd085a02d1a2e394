#!/bin/bash
# Run ON THE GPU BOX: variants of the pipelined kernel (libraries by path), small launches.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
timeout 120 python tools/queue_debug.py pipe 2>&1 | grep -v "^Traceback\|^  \|Error" | tail -12
for rep in 1 2; do
for s in ${SIZES:-4096 2048 256}; do
  for lib in "$@"; do
    AECM_LIB_PATH=$lib timeout 180 python bench.py --no-cpu-baseline --no-parity --steps 10 --warmup 2 --streams $s --blocks 2048 2>&1 | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib) S=$s', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms', d['roofline']['kernel'])"
  done
done
done
