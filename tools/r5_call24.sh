#!/bin/bash
# Run ON THE GPU BOX (round 5, call 24): front-wave priority in the sixteen-wave shape; that shape with two workgroups per CU; delay shape at two per CU.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <label> <bench args...>   (environment from the caller)
  lab=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lab $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step;', d['roofline']['launch_form'][:48])"
}
{
for rep in 1 2; do
  for s in 256 1024; do
    AECM_PIPE_GAIN=4 AECM_LIB_PATH=$L/ab_gprio3.so run fprio0 --streams $s --blocks 2048
    for v in fprio1 fprio2 fprio3; do AECM_PIPE_GAIN=4 AECM_LIB_PATH=$L/ab_$v.so run $v --streams $s --blocks 2048; done
  done
  for s in 1536 2048; do
    run base --streams $s --blocks 2048
    AECM_PIPE_GAIN=4 AECM_PIPE_DELAY=4 AECM_LIB_PATH=$L/ab_g2cu.so run g2cu --streams $s --blocks 2048
    AECM_PIPE_DELAY=4 AECM_PIPE_FRONT=2 run d4f2 --streams $s --blocks 2048
  done
done
AECM_PIPE_GAIN=4 AECM_PIPE_DELAY=4 AECM_LIB_PATH=$L/ab_g2cu.so timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --streams 2048 --blocks 2048 | tail -1 | cut -c1-300
} > $O/r5_call24.log 2>&1
cat $O/r5_call24.log
