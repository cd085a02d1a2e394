#!/bin/bash
# Run ON THE GPU BOX (round 5, call 39): seven-wave workgroups (one tail wave for four streams) with the raw hand-over and the balance, built for 8 waves per SIMD.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <label> <bench args...>   (environment from the caller)
  lab=$1; shift
  AECM_LIB_PATH=$L/ab_t1.so timeout 300 python bench.py --no-cpu-baseline --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lab $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step; parity', d['parity']['ok'], hex(d['config']['pipelined_tail_waves'] or 0), d['roofline']['launch_form'][:30])"
}
{
for rep in 1 2; do
  for s in 3584 4096; do
    run base --streams $s --blocks 2048
    AECM_PIPE_TAIL=1 run t1 --streams $s --blocks 2048
  done
done
AECM_PIPE_TAIL=1 run t1 --streams 4096 --blocks 100
} > $O/r5_call39.log 2>&1
grep -v amdgpu.ids $O/r5_call39.log
