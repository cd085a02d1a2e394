#!/bin/bash
# Run ON THE GPU BOX (round 5, call 22): the sixteen-wave shape (gain waves) at small launches; tick kernel workgroup sizes.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <label> <bench args...>   (environment from the caller)
  lab=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lab $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step; parity', d['parity']['ok'], d['roofline']['launch_form'][:48])"
}
{
( AECM_PIPE_GAIN=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or block_parity or launch_sizes" 2>&1 | tail -5 )
for rep in 1 2; do
  for s in 64 256 1024; do
    run d4 --streams $s --blocks 2048
    AECM_PIPE_GAIN=4 run g4 --streams $s --blocks 2048
  done
done
for lib in libaecm_mi355x ab_tick5 ab_tick6 ab_tick7 ab_tick14 libaecm_mi355x ab_tick7; do
  AECM_LIB_PATH=$L/$lib.so python tools/bench_sessions.py --streams 65536 --ticks 300 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib tick', d['ms_per_tick'])"
done
} > $O/r5_call22.log 2>&1
cat $O/r5_call22.log
