#!/bin/bash
# Run ON THE GPU BOX (round 5, call 33): timing probe (results wrong by construction): the six-wave shape with a barrier every second block only.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
L=webrtc_aecm_amd/_lib
run() {   # run <lib> <bench args...>
  lib=$1; shift
  AECM_LIB_PATH=$L/$lib.so timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 10 --warmup 2 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib $*', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms/step;', d['roofline']['kernel'][29:])"
}
{
for rep in 1 2 3; do
  run libaecm_mi355x --streams 4096 --blocks 2048
  run ab_halfbar --streams 4096 --blocks 2048
done
run libaecm_mi355x --streams 3584 --blocks 2048
run ab_halfbar --streams 3584 --blocks 2048
} > $O/r5_call33.log 2>&1
grep -v amdgpu.ids $O/r5_call33.log
