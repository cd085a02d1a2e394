#!/bin/bash
# Run ON THE GPU BOX (round 5, call 40): tail / delay wave priorities in the sixteen-wave shape (front waves are its longest link).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <lib> <bench args...>
  lib=$1; shift
  AECM_LIB_PATH=$L/$lib.so timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 10 --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step')"
}
{
for rep in 1 2; do
  for lib in libaecm_mi355x ab_tp0 ab_tp2 ab_dp0 ab_dp2; do
    run $lib --streams 1024 --blocks 2048
  done
done
for lib in libaecm_mi355x ab_tp0 ab_tp2; do run $lib --streams 3072 --blocks 2048; run $lib --streams 2048 --blocks 2048; done
} > $O/r5_call40.log 2>&1
grep -v amdgpu.ids $O/r5_call40.log
