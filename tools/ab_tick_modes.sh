#!/bin/bash
# Run ON THE GPU BOX: WebRtcAecmSessions tick time per tick form (AECM_TICK_MODE=flow: session machinery on the device, the default;
# lean: host-side flow classes + run-encoded one-launch tick) for uniform and per-session msInSndCardBuf, several batch sizes.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1; do
  for m in flow lean; do
    for args in "--streams 65536 --ticks 300" "--streams 65536 --ticks 300 --classes 64" "--streams 8192 --ticks 300" "--streams 1024 --ticks 300"; do
      AECM_TICK_MODE=$m timeout 180 python tools/bench_sessions.py $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', '$args', round(d['ms_per_tick'],4), 'ms/tick', d['flow_classes'])"
    done
  done
done
