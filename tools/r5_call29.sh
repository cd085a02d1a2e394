#!/bin/bash
# Run ON THE GPU BOX (round 5, call 29): tick kernel with the session's state loads issued ahead of the table fill.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
{
( AECM_LIB_PATH=$L/ab_tickearly.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streaming or session or tick" 2>&1 | tail -3 )
for rep in 1 2 3; do
  for lib in libaecm_mi355x ab_tickearly; do
    AECM_LIB_PATH=$L/$lib.so python tools/bench_sessions.py --streams 65536 --ticks 300 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib tick 65536', round(d['ms_per_tick'],4))"
    AECM_LIB_PATH=$L/$lib.so python tools/bench_sessions.py --streams 8192 --ticks 300 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib tick 8192', round(d['ms_per_tick'],4))"
  done
done
} > $O/r5_call29.log 2>&1
grep -v amdgpu.ids $O/r5_call29.log
