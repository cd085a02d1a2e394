#!/bin/bash
# Run ON THE GPU BOX (round 5, call 9): the pipelined forms as shipped (two tail waves up to 3 072 streams, balance above) by size,
# against the round-4 form (ab_base: no boost, no balance, no tails) -- and parity of the pipelined launches first.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <tail env> <lib> <bench args...>
  t=$1; lib=$2; shift; shift
  AECM_PIPE_TAIL=$t AECM_LIB_PATH=$lib timeout 200 python bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tail=$t $(basename $lib) $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step', d['roofline']['kernel'])"
}
{
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or config2 or launch_form or block_parity_vs_oracle" 2>&1 | tail -3 )
for rep in 1 2; do
  for s in 64 256 1024 2048 3072 3584 4096; do
    run 0 $L/ab_c93_b0.so --streams $s --blocks 2048
    run 2 $L/libaecm_mi355x.so --streams $s --blocks 2048
    run 2 $L/ab_tp0.so --streams $s --blocks 2048
  done
  run 2 $L/ab_g32.so --streams 4096 --blocks 2048
  run 2 $L/ab_tb0.so --streams 1024 --blocks 2048
  run 2 $L/ab_tb0.so --streams 2048 --blocks 2048
  run 2 $L/libaecm_mi355x.so --streams 4096 --blocks 2048 --fs 8000
  run 2 $L/libaecm_mi355x.so --streams 1000 --blocks 2048 --fs 8000
done
} > $O/r5_call9.log 2>&1
cat $O/r5_call9.log
