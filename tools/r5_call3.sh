#!/bin/bash
# Run ON THE GPU BOX (round 5, call 3): per-wave trace of the pipelined kernel (when workgroups finish, how long waves sit at the barrier).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
{
for s in 4096 2048 1024; do
  AECM_LIB_PATH=$L/ab_trace.so timeout 200 python tools/pipe_trace.py --streams $s --blocks 2048 2>&1 | tail -1
done
AECM_LIB_PATH=$L/ab_trace_bal.so timeout 200 python tools/pipe_trace.py --streams 4096 --blocks 2048 2>&1 | tail -1
AECM_LIB_PATH=$L/libaecm_mi355x.so timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 10 --warmup 2 --streams 4096 --blocks 2048 2>&1 | tail -1 | cut -c1-200
} > $O/r5_call3.log 2>&1
cat $O/r5_call3.log
