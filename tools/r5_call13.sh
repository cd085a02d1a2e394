#!/bin/bash
# Run ON THE GPU BOX (round 5, call 13): four front waves (one stream each) in small launches; raw hand-over without balance.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <front> <lib> <bench args...>
  f=$1; lib=$2; shift; shift
  AECM_PIPE_FRONT=$f AECM_LIB_PATH=$lib timeout 200 python bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('front=$f $(basename $lib) $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step', d['roofline']['kernel'])"
}
{
AECM_PIPE_FRONT=4 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or block_parity_vs_oracle" 2>&1 | tail -2
AECM_LIB_PATH=$L/ab_rawall.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined" 2>&1 | tail -2
for rep in 1 2 3; do
  for s in 64 256 1024 2048; do
    run 2 $L/libaecm_mi355x.so --streams $s --blocks 2048
    run 4 $L/libaecm_mi355x.so --streams $s --blocks 2048
    run 4 $L/ab_rawall.so --streams $s --blocks 2048
  done
  run 2 $L/ab_rawall.so --streams 2048 --blocks 2048
  run 2 $L/ab_rawall.so --streams 3072 --blocks 2048
  run 2 $L/libaecm_mi355x.so --streams 3072 --blocks 2048
done
} > $O/r5_call13.log 2>&1
cat $O/r5_call13.log
