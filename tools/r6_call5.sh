#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/sweep_streams.py --sizes 1024:2048:256 --blocks 2048 --set "AECM_X=0" \
  --set "AECM_PIPE_GAIN=0 AECM_PIPE_DELAY=4 AECM_PIPE_FRONT=2" --set "AECM_PIPE_DELAY=0 AECM_PIPE_FRONT=2 AECM_PIPE_RAW=1"  --set "AECM_PIPE_DELAY=0 AECM_PIPE_FRONT=2 AECM_PIPE_RAW=0" > gpurun_out/r6_sweep4.txt 2>&1
