#!/bin/bash
# Run ON THE GPU BOX (round 5, call 4): front-wave-only balance of the pipelined kernel + bulk state / session snapshot tests.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <lib> <bench args...>
  lib=$1; shift
  AECM_LIB_PATH=$lib timeout 200 python bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib) $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step', d['roofline']['kernel'])"
}
{
for rep in 1 2; do
  for v in base f2r0 f2r1 f2r1l2 f2r1p2 f2r1g4 fs1 fs2; do
    run $L/ab_$v.so --streams 4096 --blocks 2048
  done
  for v in base f2r1 f2r1g4 fs1; do
    run $L/ab_$v.so --streams 3072 --blocks 2048
    run $L/ab_$v.so --streams 2048 --blocks 2048
  done
done
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "snapshot or bulk_state or session_migrates" 2>&1 | tail -15 )
} > $O/r5_call4.log 2>&1
cat $O/r5_call4.log
