#!/bin/bash
# Run ON THE GPU BOX (round 5, call 1): progress-feedback balance of the pipelined kernel against the round-4 form, its
# parameters, and the cost of release / acquire on the chunk queue's hand-over flag.  Interleaved repetitions.
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
# parity of the new default first (pipelined launches + the queue)
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or launch_form or config2" 2>&1 | tail -5 )
for rep in 1 2; do
  for lib in $L/ab_base.so $L/libaecm_mi355x.so $L/ab_bal_d1.so $L/ab_bal_d2.so $L/ab_bal_g4.so $L/ab_bal_g16.so $L/ab_bal_lead2.so $L/ab_bal_f1.so; do
    run $lib --streams 4096 --blocks 2048
  done
  for lib in $L/ab_base.so $L/libaecm_mi355x.so $L/ab_bal_d1.so $L/ab_bal_g16.so $L/ab_bal_f1.so; do
    run $lib --streams 2048 --blocks 2048
  done
  for lib in $L/ab_base.so $L/libaecm_mi355x.so; do
    run $lib --streams 1024 --blocks 2048
    run $lib --streams 3072 --blocks 2048
  done
  for lib in $L/libaecm_mi355x.so $L/ab_fence1.so; do
    STEPS=6 run $lib
    STEPS=6 run $lib --streams 8192
  done
done
} > $O/r5_call1.log 2>&1
cat $O/r5_call1.log
