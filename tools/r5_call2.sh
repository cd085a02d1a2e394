#!/bin/bash
# Run ON THE GPU BOX (round 5, call 2): progress-feedback balance of the pipelined kernel, second form (per-workgroup progress
# words, monitor in a front wave), against the round-4 form.  Interleaved repetitions.
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
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or launch_form or config2 or contention" 2>&1 | tail -5 )
for rep in 1 2; do
  for v in base libaecm_mi355x bal_dry bal_d1 bal_d2 bal_g4 bal_g16 bal_g32 bal_lead0 bal_lead2 bal_f1; do
    lib=$L/ab_$v.so; [ $v = libaecm_mi355x ] && lib=$L/libaecm_mi355x.so
    run $lib --streams 4096 --blocks 2048
  done
  for v in base libaecm_mi355x bal_dry bal_d1 bal_g16 bal_lead0; do
    lib=$L/ab_$v.so; [ $v = libaecm_mi355x ] && lib=$L/libaecm_mi355x.so
    run $lib --streams 2048 --blocks 2048
    run $lib --streams 3072 --blocks 2048
  done
  for v in base libaecm_mi355x; do
    lib=$L/ab_$v.so; [ $v = libaecm_mi355x ] && lib=$L/libaecm_mi355x.so
    run $lib --streams 1024 --blocks 2048
  done
done
} > $O/r5_call2.log 2>&1
cat $O/r5_call2.log
