#!/bin/bash
# Run ON THE GPU BOX (round 5, call 18): packed gain product / comfort-noise add (hot path 581 -> 577 VALU) against the library before it.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <lib> <bench args...>
  lib=$1; shift
  AECM_LIB_PATH=$lib timeout 300 python bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-10} --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib) $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step')"
}
{
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 )
for rep in 1 2 3; do
  for lib in $L/ab_before_pk.so $L/libaecm_mi355x.so; do
    run $lib
    run $lib --streams 4096 --blocks 2048
    run $lib --clean
    AECM_LIB_PATH=$lib python tools/bench_sessions.py --streams 65536 --ticks 300 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib) tick', d['ms_per_tick'])"
  done
done
} > $O/r5_call18.log 2>&1
cat $O/r5_call18.log
