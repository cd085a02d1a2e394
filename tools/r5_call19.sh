#!/bin/bash
# Run ON THE GPU BOX (round 5, call 19): the delay estimator in waves of its own (AECM_PIPE_DELAY=4), parity and rates at small launches.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
run() {   # run <label> <bench args...>   (environment from the caller)
  lab=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lab $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step; parity', d['parity']['ok'], d['roofline']['launch_form'][:48])"
}
{
( AECM_PIPE_DELAY=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or block_parity or launch_sizes" 2>&1 | tail -5 )
( AECM_PIPE_DELAY=4 AECM_PIPE_FRONT=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined_launch_sizes" 2>&1 | tail -3 )
for rep in 1 2; do
  for s in 64 256 1024 2048; do
    AECM_PIPE_DELAY=0 run base --streams $s --blocks 2048
    AECM_PIPE_DELAY=4 AECM_PIPE_FRONT=2 run d4f2 --streams $s --blocks 2048
    AECM_PIPE_DELAY=4 AECM_PIPE_FRONT=4 run d4f4 --streams $s --blocks 2048
  done
done
} > $O/r5_call19.log 2>&1
cat $O/r5_call19.log
