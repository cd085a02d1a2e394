#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=webrtc_aecm_amd/_lib
for lib in libaecm_mi355x ab_norot ab_rot2; do
echo "== $lib" >> gpurun_out/r6_sweep7.txt
AECM_LIB_PATH=$L/$lib.so python tools/sweep_streams.py --sizes 256:4096:256 --blocks 2048 --set "AECM_X=0" \
  --set "AECM_PIPE_DELAY=2 AECM_PIPE_GAIN=4 AECM_PIPE_FRONT=4" >> gpurun_out/r6_sweep7.txt 2>&1
done
