#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export AECM_LIB_PATH=webrtc_aecm_amd/_lib/ab_trace.so
for s in 1024 1280 2048; do
AECM_PIPE_DELAY=2 AECM_PIPE_GAIN=4 AECM_PIPE_FRONT=4 python tools/pipe_trace.py --streams $s --blocks 512 2>&1 | tail -1 > gpurun_out/r6_trace_$s.json
done
python tools/pipe_trace.py --streams 1536 --blocks 512 2>&1 | tail -1 > gpurun_out/r6_trace_1536_10w.json
python tools/pipe_trace.py --streams 2560 --blocks 512 2>&1 | tail -1 > gpurun_out/r6_trace_2560_8w.json
