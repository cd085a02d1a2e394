#!/bin/bash
# Run ON THE GPU BOX (round 5, call 37): a twelve-wave shape with gain waves but no delay waves (4 channel + 2 front + 2 tail + 4 gain) at two workgroups per CU.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <label> <bench args...>   (environment from the caller)
  lab=$1; shift
  AECM_LIB_PATH=$L/ab_g12.so timeout 300 python bench.py --no-cpu-baseline --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lab $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step; parity', d['parity']['ok'], hex(d['config']['pipelined_tail_waves'] or 0), d['roofline']['launch_form'][:30])"
}
{
( AECM_LIB_PATH=$L/ab_g12.so AECM_PIPE_DELAY=0 AECM_PIPE_GAIN=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined_launch_sizes or block_parity" 2>&1 | tail -3 )
for rep in 1 2; do
  for s in 1024 1536 2048; do
    run base --streams $s --blocks 2048
    AECM_PIPE_DELAY=0 AECM_PIPE_GAIN=4 run g12 --streams $s --blocks 2048
  done
done
} > $O/r5_call37.log 2>&1
grep -v amdgpu.ids $O/r5_call37.log
