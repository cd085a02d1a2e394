#!/bin/bash
# Run ON THE GPU BOX: the round's last check of the library as committed -- the whole GPU suite, the default bench line (its issue_bound /
# traffic quoted from the committed rocprofv3 records of the same kernels), the tick kernel's rocprofv3 record and serving-path numbers.
#   gpurun --timeout 1500 -- 'bash tools/gpu_round5_verify.sh'     then copy gpurun_out/tickonly_tick/tick_kernel_stats.csv to profiles/r05_tick_kernel_stats.csv
set -u
O=gpurun_out
mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/r5v_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r5v_pytest.log
python bench.py --steps 20 --warmup 5 > $O/r5v_bench.log 2>&1
PREFIX=tickonly PARTS="tick" bash tools/profile_gpu.sh > $O/r5v_profile_tick.log 2>&1
{
python tools/bench_sessions.py --streams 65536 --ticks 300 | tail -1
python tools/bench_sessions.py --streams 65536 --ticks 300 --async | tail -1
python tools/bench_sessions.py --streams 65536 --fs 8000 --ticks 300 | tail -1
for s in 1024 8192; do python tools/bench_sessions.py --streams $s --ticks 300 | tail -1; done
python tools/bench_single_session.py | tail -1
} > $O/r5v_sessions.log 2>&1
tail -4 $O/r5v_pytest.log; tail -1 $O/r5v_bench.log | cut -c1-300; grep -v amdgpu.ids $O/r5v_sessions.log | cut -c1-260
python -c "
import json
d = json.loads(open('$O/r5v_bench.log').read().strip().splitlines()[-1])
print('bench', round(d['value']/1e6,1), 'parity', d['parity']['ok'], 'issue_bound available', d['roofline']['issue_bound'].get('available'), 'traffic', d['roofline']['traffic'], 'commit', d['config']['commit'])"
