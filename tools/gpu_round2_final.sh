#!/bin/bash
# Run ON THE GPU BOX: the round's closing measurements (tests, bench line, config sweep, profiles, serving-path numbers).
set -u
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r2f_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2f_pytest.log
python bench.py > gpurun_out/r2f_bench.log 2>&1
PARTS="stats hbm sq cal tick" tools/profile_gpu.sh > gpurun_out/r2f_profile.log 2>&1
{
for a in "--streams 4096 --blocks 2048" "--fs 8000 --streams 32768" "--streams 131072 --blocks 512" "--clean" "--streams 16384"; do
  python bench.py --no-cpu-baseline $a | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', round(d['value']/1e6,1), 'M frames/s;', d['config']['workload'][:70])"
done
python tools/bench_host_io.py 2>&1 | tail -1
python tools/bench_single_session.py | tail -1
for c in 1 16 256; do python tools/bench_sessions.py --streams 65536 --ticks 300 --classes $c | tail -1; done
for s in 1024 8192; do python tools/bench_sessions.py --streams $s --ticks 300 | tail -1; done
} > gpurun_out/r2f_sweep.log 2>&1
tail -3 gpurun_out/r2f_pytest.log; tail -1 gpurun_out/r2f_bench.log | cut -c1-300; cat gpurun_out/r2f_sweep.log
