#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r6_gpu_full2.log
python tools/sweep_streams.py --sizes 256:8192:256 --blocks 2048 > gpurun_out/r6_sweep10.txt 2>&1
python tools/sweep_streams.py --sizes 1792,1920,2048 --blocks 2048 --set "pipe_rot=-1" --set "pipe_delay_waves=0 pipe_front_waves=4 pipe_raw=1" >> gpurun_out/r6_sweep10.txt 2>&1
for s in 1536 2560 1280 768; do python tools/soak_parity.py --streams $s --blocks 1280 --passes 2 2>&1 | tail -1 >> gpurun_out/r6_soak.jsonl; done
