#!/bin/bash
# Run ON THE GPU BOX: tools/bench_sessions.py for A/B library variants.   tools/ab_sessions.sh variant...   (STREAMS from env)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
for v in "$@"; do
  for s in ${STREAMS:-65536 8192}; do
    AECM_LIB_PATH=webrtc_aecm_amd/_lib/ab_$v.so timeout 180 python tools/bench_sessions.py --streams $s 2>&1 | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['streams'], round(d['ms_per_tick'],4), 'ms/tick', d['realtime_streams_per_gpu'])"
  done
done
