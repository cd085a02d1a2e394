#!/bin/bash
# Run ON THE GPU BOX (round 5, call 41): the GPU suite, smoke and the default bench line after the engine's launch-form refactor.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
{
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 )
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 | tail -1 > $O/r5_call41_bench_line.json
python -c "
import json
d = json.load(open('$O/r5_call41_bench_line.json'))
print('bench', round(d['value']/1e6,1), 'parity', d['parity']['ok'], 'issue_bound available', d['roofline']['issue_bound'].get('available'), 'traffic', d['roofline']['traffic'], 'commit', d['config']['commit'])"
} > $O/r5_call41.log 2>&1
grep -v amdgpu.ids $O/r5_call41.log
