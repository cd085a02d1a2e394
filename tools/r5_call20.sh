#!/bin/bash
# Run ON THE GPU BOX (round 5, call 20): delay waves that also fetch the aligned far-history row for the middle wave; traces; delay-wave priority.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <label> <bench args...>   (environment from the caller)
  lab=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lab $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step; parity', d['parity']['ok'], d['roofline']['launch_form'][:48])"
}
{
( AECM_PIPE_DELAY=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or block_parity or launch_sizes" 2>&1 | tail -5 )
for rep in 1 2; do
  for s in 64 256 1024; do
    AECM_PIPE_DELAY=0 run base --streams $s --blocks 2048
    AECM_PIPE_DELAY=4 run d4 --streams $s --blocks 2048
    AECM_PIPE_DELAY=4 AECM_LIB_PATH=$L/ab_dprio0.so run d4prio0 --streams $s --blocks 2048
    AECM_PIPE_DELAY=4 AECM_LIB_PATH=$L/ab_dprio2.so run d4prio2 --streams $s --blocks 2048
  done
  AECM_PIPE_DELAY=4 run d4 --streams 1536 --blocks 2048
  AECM_PIPE_DELAY=0 run base --streams 1536 --blocks 2048
done
for s in 1024 256; do
  AECM_PIPE_DELAY=4 AECM_LIB_PATH=$L/ab_trace.so python tools/pipe_trace.py --streams $s --blocks 2048 2>&1 | tail -1
  AECM_PIPE_DELAY=4 AECM_PIPE_FRONT=4 AECM_LIB_PATH=$L/ab_trace.so python tools/pipe_trace.py --streams $s --blocks 2048 2>&1 | tail -1
done
} > $O/r5_call20.log 2>&1
cat $O/r5_call20.log
