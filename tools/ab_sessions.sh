#!/bin/bash
# Run ON THE GPU BOX: tools/bench_sessions.py for A/B library variants (webrtc_aecm_amd/_lib/ab_<v>.so), REPS interleaved repetitions.
#   tools/ab_sessions.sh variant1 variant2 ...       (SESS_ARGS / REPS from the environment)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    AECM_LIB_PATH=webrtc_aecm_amd/_lib/ab_$v.so timeout 180 python tools/bench_sessions.py ${SESS_ARGS:---streams 65536 --ticks 300} 2>&1 | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_tick'],4), 'ms/tick')"
  done
done
