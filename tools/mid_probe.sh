#!/bin/bash
# Run ON THE GPU BOX: which launch form wins below the chip's resident wave count (experiments: AECM_QUEUE_MIN_STREAMS lowers the
# chunk queue's threshold, AECM_PIPELINED=0 turns the pipelined form off) -- profiles/r04_experiments.md section 4.
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env "$1" "$2" "$3" timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 10 --warmup 2 --streams $4 --blocks ${BLOCKS:-1280} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 $3 S=$4', round(d['value']/1e6,1), 'M', d['roofline']['kernel'])"; }
for rep in 1 2; do
for s in ${SIZES:-1024 2048 3072 4096}; do
  run AECM_X=0 AECM_Y=0 AECM_Z=0 $s
  run AECM_QUEUE_MIN_STREAMS=0 AECM_PIPELINED=0 AECM_QUEUE_CHUNK=128 $s
  run AECM_QUEUE_MIN_STREAMS=0 AECM_PIPELINED=0 AECM_QUEUE_CHUNK=32 $s
  run AECM_QUEUE_MIN_STREAMS=0 AECM_PIPELINED=0 AECM_QUEUE_CHUNK=16 $s
done
done
