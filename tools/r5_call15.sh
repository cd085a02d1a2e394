#!/bin/bash
# Run ON THE GPU BOX (round 5, call 15): the delay estimator in the front waves (small pipelined shapes): parity, then rates.
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
AECM_PIPE_DE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or config2 or block_parity_vs_oracle or state_snapshot" 2>&1 | tail -2
AECM_PIPE_DE=1 AECM_PIPE_FRONT=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or block_parity_vs_oracle" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or launch_form or block_parity_vs_oracle" 2>&1 | tail -2
for rep in 1 2 3; do
  for s in 64 256 1024; do
    run AECM_PIPE_DE=0 --streams $s --blocks 2048
    run AECM_PIPE_DE=1 --streams $s --blocks 2048
  done
  run AECM_PIPE_DE=0 --streams 2048 --blocks 2048
  run AECM_PIPE_DE=1 --streams 2048 --blocks 2048
  run "AECM_PIPE_DE=1 AECM_PIPE_FRONT=2" --streams 2048 --blocks 2048
  run AECM_PIPE_DE=0 --streams 1536 --blocks 2048
  run AECM_PIPE_DE=1 --streams 1536 --blocks 2048
  run AECM_PIPE_DE=0 --streams 3072 --blocks 2048
  run AECM_PIPE_DE=1 --streams 3072 --blocks 2048
done
} > $O/r5_call15.log 2>&1
cat $O/r5_call15.log
