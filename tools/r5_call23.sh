#!/bin/bash
# Run ON THE GPU BOX (round 5, call 23): trace of the sixteen-wave shape; the gain waves' priority.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <label> <bench args...>   (environment from the caller)
  lab=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lab $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step;', d['roofline']['launch_form'][:48])"
}
{
for s in 1024 256; do
  AECM_PIPE_GAIN=4 AECM_LIB_PATH=$L/ab_trace.so python tools/pipe_trace.py --streams $s --blocks 2048 2>&1 | tail -1
done
for rep in 1 2; do
  for s in 256 1024; do
    AECM_PIPE_GAIN=4 run g4prio2 --streams $s --blocks 2048
    AECM_PIPE_GAIN=4 AECM_LIB_PATH=$L/ab_gprio1.so run g4prio1 --streams $s --blocks 2048
    AECM_PIPE_GAIN=4 AECM_LIB_PATH=$L/ab_gprio3.so run g4prio3 --streams $s --blocks 2048
  done
done
} > $O/r5_call23.log 2>&1
cat $O/r5_call23.log
