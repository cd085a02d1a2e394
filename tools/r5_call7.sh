#!/bin/bash
# Run ON THE GPU BOX (round 5, call 7): the tail role of the pipelined kernel (0 / 1 / 2 tail waves per workgroup) by size; parity first.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <tail> <lib> <bench args...>
  t=$1; lib=$2; shift; shift
  AECM_PIPE_TAIL=$t AECM_LIB_PATH=$lib timeout 200 python bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tail=$t $(basename $lib) $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step', d['roofline']['kernel'])"
}
{
for t in 1 2; do
  ( AECM_PIPE_TAIL=$t timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or config2 or block_parity_vs_oracle" 2>&1 | tail -4 )
done
for rep in 1 2; do
  for s in 4096 3072 2048 1024 256; do
    for t in 0 1 2; do
      run $t $L/libaecm_mi355x.so --streams $s --blocks 2048
    done
  done
  for v in tp0 tp2 tp3; do
    run 1 $L/ab_$v.so --streams 4096 --blocks 2048
    run 2 $L/ab_$v.so --streams 2048 --blocks 2048
  done
done
for t in 1 2; do
  AECM_PIPE_TAIL=$t AECM_LIB_PATH=$L/ab_trace.so timeout 200 python tools/pipe_trace.py --streams $((t == 1 ? 4096 : 2048)) --blocks 2048 2>&1 | tail -1
done
} > $O/r5_call7.log 2>&1
cat $O/r5_call7.log
