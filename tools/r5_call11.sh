#!/bin/bash
# Run ON THE GPU BOX (round 5, call 11): raw hand-over (spectra formed by the back waves) in the balanced pipelined kernel; boost on / off there.
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
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or config2 or launch_form" 2>&1 | tail -3 )
for rep in 1 2 3; do
  for v in libaecm_mi355x raw0 raw1b0 raw0b0; do
    lib=$L/ab_$v.so; [ $v = libaecm_mi355x ] && lib=$L/libaecm_mi355x.so
    run $lib --streams 4096 --blocks 2048
    run $lib --streams 3584 --blocks 2048
  done
done
} > $O/r5_call11.log 2>&1
cat $O/r5_call11.log
