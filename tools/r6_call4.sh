#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=webrtc_aecm_amd/_lib
python tools/sweep_streams.py --sizes 256:2048:256 --blocks 2048 --set "AECM_X=0" \
  --set "AECM_PIPE_WGS=1" --set "AECM_PIPE_DELAY=2 AECM_PIPE_GAIN=4 AECM_PIPE_FRONT=4" > gpurun_out/r6_sweep3a.txt 2>&1
AECM_LIB_PATH=$L/ab_p10.so python tools/sweep_streams.py --sizes 1024:3072:256 --blocks 2048 --set "AECM_X=0" \
  --set "AECM_PIPE_DELAY=0 AECM_PIPE_FRONT=4 AECM_PIPE_RAW=1" > gpurun_out/r6_sweep3b.txt 2>&1
AECM_LIB_PATH=$L/ab_p8.so python tools/sweep_streams.py --sizes 2048:4096:256 --blocks 2048 --set "AECM_X=0" \
  --set "AECM_PIPE_DELAY=0 AECM_PIPE_FRONT=2 AECM_PIPE_RAW=1 AECM_PIPE_TAIL=2" > gpurun_out/r6_sweep3c.txt 2>&1
