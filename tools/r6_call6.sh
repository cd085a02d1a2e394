#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/sweep_streams.py --sizes 1024,1028,1032,1056,1088,1152,1216,1280,1408,1536 --blocks 2048 \
  --set "AECM_PIPE_DELAY=2 AECM_PIPE_GAIN=4 AECM_PIPE_FRONT=4" --set "AECM_PIPE_DELAY=2 AECM_PIPE_GAIN=4 AECM_PIPE_FRONT=4 AECM_PIPE_SPREAD=0" > gpurun_out/r6_sweep5.txt 2>&1
python tools/sweep_streams.py --sizes 512,516,520,576,640,768 --blocks 2048 \
  --set "AECM_PIPE_WGS=1" --set "AECM_PIPE_WGS=2" >> gpurun_out/r6_sweep5.txt 2>&1
