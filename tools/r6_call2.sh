#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests -m gpu -x -q -k "pipelined" 2>&1 | tail -15 > gpurun_out/r6_pipe_tests.log
python tools/sweep_streams.py --sizes 256:4096:256 --blocks 2048 --set "AECM_PIPE_SPREAD=1" --set "AECM_PIPE_SPREAD=0" > gpurun_out/r6_sweep1.txt 2>&1
