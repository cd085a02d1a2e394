#!/bin/bash
# Run ON THE GPU BOX (round 5, call 14): the pipelined shapes as shipped (by size) + raw hand-over at one workgroup per CU; parity; traces.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <env assignments or -> <bench args...>
  e=$1; shift
  env $e AECM_LIB_PATH=$L/libaecm_mi355x.so timeout 200 python bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step', d['roofline']['kernel'])"
}
{
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or config2 or launch_form or block_parity_vs_oracle" 2>&1 | tail -2
for rep in 1 2 3; do
  for s in 64 256 1024 1536 2048 2560 3072 3584 4096; do run X=0 --streams $s --blocks 2048; done
  for s in 256 1024; do run AECM_PIPE_RAW=1 --streams $s --blocks 2048; done
  run AECM_PIPE_FRONT=2 --streams 1536 --blocks 2048
  run AECM_PIPE_FRONT=4 --streams 2560 --blocks 2048
  run X=0 --streams 4096 --blocks 2048 --fs 8000
  run X=0 --streams 1000 --blocks 2048 --fs 8000
done
for s in 1024 2048 3072 4096; do AECM_LIB_PATH=$L/ab_trace.so timeout 200 python tools/pipe_trace.py --streams $s --blocks 2048 2>&1 | tail -1; done
} > $O/r5_call14.log 2>&1
cat $O/r5_call14.log
