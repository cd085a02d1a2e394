#!/bin/bash
# Run ON THE GPU BOX (round 5, call 35): persistent tick kernel with one claim counter per XCD and batched claims.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
t() { python tools/bench_sessions.py "$@" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_tick'],4))"; }
{
( AECM_TICK_PERSISTENT=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streaming or session or tick" 2>&1 | tail -2 )
for rep in 1 2 3; do
  echo "one-shot waves 65536: $(t --streams 65536 --ticks 300)"
  echo "persistent, claims of 4: $(AECM_TICK_PERSISTENT=1 t --streams 65536 --ticks 300)"
  echo "persistent, claims of 2: $(AECM_TICK_PERSISTENT=1 AECM_LIB_PATH=$L/ab_claim2.so t --streams 65536 --ticks 300)"
  echo "persistent, claims of 8: $(AECM_TICK_PERSISTENT=1 AECM_LIB_PATH=$L/ab_claim8.so t --streams 65536 --ticks 300)"
done
echo "8 kHz one-shot: $(t --streams 65536 --fs 8000 --ticks 300)  persistent: $(AECM_TICK_PERSISTENT=1 t --streams 65536 --fs 8000 --ticks 300)"
echo "16384 one-shot: $(t --streams 16384 --ticks 300)  persistent: $(AECM_TICK_PERSISTENT=1 t --streams 16384 --ticks 300)"
} > $O/r5_call35.log 2>&1
grep -v amdgpu.ids $O/r5_call35.log
