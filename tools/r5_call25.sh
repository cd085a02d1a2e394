#!/bin/bash
# Run ON THE GPU BOX (round 5, call 25): the whole GPU suite with the sixteen-wave shape as the small launches' form; every frame of such launches against the reference.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
{
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 )
for s in 64 1024; do python tools/soak_parity.py --streams $s --blocks 2048 2>&1 | tail -1 | cut -c1-400; done
for s in 4 64 256 512 1024; do
  python bench.py --no-cpu-baseline --streams $s --blocks 2048 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$s streams', round(d['value']/1e6,1), 'M frames/s; parity', d['parity']['ok'], ';', d['roofline']['kernel'])"
done
} > $O/r5_call25.log 2>&1
cat $O/r5_call25.log
