#!/bin/bash
# Run ON THE GPU BOX (round 5, call 30): timing probes of the tick kernel (results wrong by construction): what its two store -> fence -> load round trips cost.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
{
for rep in 1 2; do
  for lib in ab_early1 ab_probe1 ab_probe2 ab_probe3; do
    AECM_LIB_PATH=$L/$lib.so python tools/bench_sessions.py --streams 65536 --ticks 300 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib tick 65536', round(d['ms_per_tick'],4))"
  done
done
} > $O/r5_call30.log 2>&1
grep -v amdgpu.ids $O/r5_call30.log
