#!/bin/bash
# Run ON THE GPU BOX (round 5, call 8): (a) what the runtime-gated / restructured pipelined kernel lost against the variants of
# calls 5-6, (b) the tail role with seven-wave workgroups built for 8 waves per SIMD.
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
( AECM_PIPE_TAIL=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or config2" 2>&1 | tail -3 )
for rep in 1 2; do
  for s in 4096 3072 2048 1024; do
    for v in c93 c93_b0 cur_b0 cur_boost0 libaecm_mi355x; do
      lib=$L/ab_$v.so; [ $v = libaecm_mi355x ] && lib=$L/libaecm_mi355x.so
      run 0 $lib --streams $s --blocks 2048
    done
  done
  for s in 4096 3584 3072 2048; do
    run 1 $L/libaecm_mi355x.so --streams $s --blocks 2048
  done
done
AECM_PIPE_TAIL=1 AECM_LIB_PATH=$L/ab_trace.so timeout 200 python tools/pipe_trace.py --streams 4096 --blocks 2048 2>&1 | tail -1
} > $O/r5_call8.log 2>&1
cat $O/r5_call8.log
