#!/bin/bash
# Run ON THE GPU BOX (round 5, call 12): raw hand-over in the instantiations without balance too?
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
for rep in 1 2 3; do
  for v in libaecm_mi355x rawall; do
    lib=$L/ab_$v.so; [ $v = libaecm_mi355x ] && lib=$L/libaecm_mi355x.so
    for s in 256 1024 2048 3072 4096; do run $lib --streams $s --blocks 2048; done
  done
done
AECM_LIB_PATH=$L/ab_rawall.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined" 2>&1 | tail -2
} > $O/r5_call12.log 2>&1
cat $O/r5_call12.log
