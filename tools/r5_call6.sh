#!/bin/bash
# Run ON THE GPU BOX (round 5, call 6): group length / allowed lead / sampled monitor of the front-wave balance; trace of the best.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <lib> <bench args...>
  lib=$1; shift
  AECM_LIB_PATH=$lib timeout 200 python bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib) $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step')"
}
{
for rep in 1 2; do
  for v in base g16 g16l0 g32 g32l0 g64l0 g16s g8s g16l2 g16m3; do
    run $L/ab_$v.so --streams 4096 --blocks 2048
  done
  for v in base g16 g16s g32l0; do
    run $L/ab_$v.so --streams 3584 --blocks 2048
    run $L/ab_$v.so --streams 4096 --blocks 2048 --fs 8000
  done
done
AECM_LIB_PATH=$L/ab_trace_g16.so timeout 200 python tools/pipe_trace.py --streams 4096 --blocks 2048 2>&1 | tail -1
} > $O/r5_call6.log 2>&1
cat $O/r5_call6.log
