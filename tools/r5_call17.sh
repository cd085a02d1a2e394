#!/bin/bash
# Run ON THE GPU BOX (round 5, call 17): the balance in the two-tail shape (three workgroups per CU).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <env assignments> <bench args...>
  e=$1; shift
  env $e AECM_LIB_PATH=$L/libaecm_mi355x.so timeout 200 python bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step', d['roofline']['kernel'])"
}
{
AECM_PIPE_BALANCE_TAILS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or config2" 2>&1 | tail -2
for rep in 1 2 3; do
  for s in 2560 2816 3072; do
    run AECM_PIPE_BALANCE_TAILS=0 --streams $s --blocks 2048
    run AECM_PIPE_BALANCE_TAILS=1 --streams $s --blocks 2048
  done
done
} > $O/r5_call17.log 2>&1
cat $O/r5_call17.log
