#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests -m gpu -x -q -k "pipelined" 2>&1 | tail -5 > gpurun_out/r6_pipe_tests2.log
python tools/sweep_streams.py --sizes 256:4096:256 --blocks 2048 --set "AECM_X=0" \
  --set "AECM_PIPE_DELAY=2 AECM_PIPE_GAIN=4 AECM_PIPE_FRONT=4" > gpurun_out/r6_sweep6.txt 2>&1
python tools/sweep_streams.py --sizes 1028,1032,1088 --blocks 2048 \
  --set "AECM_PIPE_DELAY=2 AECM_PIPE_GAIN=4 AECM_PIPE_FRONT=4" >> gpurun_out/r6_sweep6.txt 2>&1
