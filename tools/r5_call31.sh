#!/bin/bash
# Run ON THE GPU BOX (round 5, call 31): short launches (1 .. 32 blocks) of small batches: pipelined shapes against one wavefront per stream; smoke().
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
run() {   # run <label> <bench args...>
  lab=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 200 --warmup 20 "$@" 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lab $*', round(d['ms_per_step']*1000,2), 'us per launch (wall)', round(d['roofline']['kernel_avg_ms']*1000,2), 'us kernel;', d['roofline']['kernel'][:60])"
}
{
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for s in 256 1024 4096; do
  for t in 1 2 3 4 6 8 16 32; do
    run pipe --streams $s --blocks $t
    AECM_PIPELINED=0 run wave --streams $s --blocks $t
  done
done
} > $O/r5_call31.log 2>&1
grep -v amdgpu.ids $O/r5_call31.log
