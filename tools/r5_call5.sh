#!/bin/bash
# Run ON THE GPU BOX (round 5, call 5): front-wave priority control of the pipelined kernel (levels, progressive boost, proportional feedback).
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
  for v in base fs1 f2r0 f2r0b f2r0_l0 f2r0_g16 f2r0_p2 f3 f3b f3_l0 f3_l0b f3_p3 f3_g4; do
    run $L/ab_$v.so --streams 4096 --blocks 2048
  done
  for v in base fs1 f2r0b f3b f3_l0b; do
    run $L/ab_$v.so --streams 3072 --blocks 2048
    run $L/ab_$v.so --streams 2048 --blocks 2048
    run $L/ab_$v.so --streams 1024 --blocks 2048
  done
done
} > $O/r5_call5.log 2>&1
cat $O/r5_call5.log
