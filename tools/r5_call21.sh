#!/bin/bash
# Run ON THE GPU BOX (round 5, call 21): where the waves of the pipelined shapes land (CU / SIMD by role), front 2 vs 4 with delay waves.
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
for s in 4096 3072 2048 1024; do
  AECM_LIB_PATH=$L/ab_trace.so python tools/pipe_trace.py --streams $s --blocks 2048 2>&1 | tail -1
done
for rep in 1 2; do
  for s in 64 256 1024; do
    run d4f2 --streams $s --blocks 2048
    AECM_PIPE_FRONT=4 run d4f4 --streams $s --blocks 2048
  done
done
} > $O/r5_call21.log 2>&1
cat $O/r5_call21.log
