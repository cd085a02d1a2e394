#!/bin/bash
# Run ON THE GPU BOX (round 5, call 27): a thirteen-wave shape (4 channel + 2 front + 2 tail + 1 delay + 4 gain) at two workgroups per CU.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <label> <bench args...>   (environment from the caller)
  lab=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lab $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step; parity', d['parity']['ok'], d['config']['pipelined_tail_waves'], d['roofline']['launch_form'][:30])"
}
{
for rep in 1 2; do
  for s in 1024 1536 2048; do
    AECM_LIB_PATH=$L/ab_x13.so run base --streams $s --blocks 2048
    AECM_PIPE_DELAY=1 AECM_PIPE_GAIN=4 AECM_LIB_PATH=$L/ab_x13.so run x13 --streams $s --blocks 2048
  done
done
} > $O/r5_call27.log 2>&1
cat $O/r5_call27.log
