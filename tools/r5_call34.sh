#!/bin/bash
# Run ON THE GPU BOX (round 5, call 34): the tick kernel as a persistent grid claiming sessions from a counter (AECM_TICK_PERSISTENT).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
{
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streaming or session or tick" 2>&1 | tail -2 )
( AECM_TICK_PERSISTENT=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streaming or session or tick" 2>&1 | tail -2 )
for rep in 1 2 3; do
  python tools/bench_sessions.py --streams 65536 --ticks 300 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-shot waves   65536', round(d['ms_per_tick'],4))"
  AECM_TICK_PERSISTENT=1 python tools/bench_sessions.py --streams 65536 --ticks 300 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persistent x1    65536', round(d['ms_per_tick'],4))"
  AECM_TICK_PERSISTENT=2 python tools/bench_sessions.py --streams 65536 --ticks 300 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persistent x2    65536', round(d['ms_per_tick'],4))"
done
python tools/bench_sessions.py --streams 65536 --fs 8000 --ticks 300 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-shot waves   8 kHz', round(d['ms_per_tick'],4))"
AECM_TICK_PERSISTENT=1 python tools/bench_sessions.py --streams 65536 --fs 8000 --ticks 300 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persistent x1    8 kHz', round(d['ms_per_tick'],4))"
python tools/bench_sessions.py --streams 16384 --ticks 300 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-shot waves   16384', round(d['ms_per_tick'],4))"
AECM_TICK_PERSISTENT=1 python tools/bench_sessions.py --streams 16384 --ticks 300 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persistent x1    16384', round(d['ms_per_tick'],4))"
} > $O/r5_call34.log 2>&1
grep -v amdgpu.ids $O/r5_call34.log
