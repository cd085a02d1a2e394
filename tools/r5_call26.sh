#!/bin/bash
# Run ON THE GPU BOX (round 5, call 26): delay waves in the mid-size shapes (one per workgroup at three / four workgroups per CU, two next to four front waves at two per CU).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
run() {   # run <label> <bench args...>   (environment from the caller)
  lab=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lab $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step; parity', d['parity']['ok'], d['roofline']['kernel'][29:])"
}
{
( AECM_PIPE_DELAY=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined_launch_sizes or block_parity" 2>&1 | tail -3 )
for rep in 1 2; do
  run base --streams 2048 --blocks 2048
  AECM_PIPE_DELAY=2 run d2f4 --streams 2048 --blocks 2048
  run base --streams 3072 --blocks 2048
  AECM_PIPE_DELAY=1 run d1 --streams 3072 --blocks 2048
  run base --streams 3584 --blocks 2048
  AECM_PIPE_DELAY=1 run d1 --streams 3584 --blocks 2048
  run base --streams 4096 --blocks 2048
  AECM_PIPE_DELAY=1 run d1 --streams 4096 --blocks 2048
done
} > $O/r5_call26.log 2>&1
cat $O/r5_call26.log
