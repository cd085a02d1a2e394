#!/bin/bash
# Run ON THE GPU BOX: the round's closing measurements (tests, bench line, profiles, config sweep, serving-path numbers).
#   gpurun --timeout 2400 -- 'bash tools/gpu_round4_final.sh'      then: python tools/summarize_profile.py r04
set -u
O=gpurun_out
mkdir -p $O
( time AECM_SANITIZER_LOG=$PWD/$O/r4f_ubsan_gpu.log python -m pytest tests -m gpu -x -q --durations=6 ) > $O/r4f_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r4f_pytest.log
python bench.py --steps 20 --warmup 5 > $O/r4f_bench.log 2>&1
PARTS="stats hbm sq cal tick" bash tools/profile_gpu.sh > $O/r4f_profile.log 2>&1
{
for a in "--streams 4096 --blocks 2048" "--streams 16384" "--fs 8000 --streams 32768" "--streams 131072 --blocks 512" "--clean" "--variant safe" "--streams 1024 --blocks 2048"; do
  python bench.py --no-cpu-baseline $a | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', round(d['value']/1e6,1), 'M frames/s; parity', d['parity']['ok'], ';', d['roofline']['kernel'], ';', d['config']['workload'][:70])"
done
AECM_QUEUE_CHUNK=0 python bench.py --no-cpu-baseline --no-parity | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one wave per stream (AECM_QUEUE_CHUNK=0)', round(d['value']/1e6,1), 'M frames/s;', d['roofline']['kernel'])"
AECM_PIPELINED=0 python bench.py --no-cpu-baseline --no-parity --streams 4096 --blocks 2048 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4096 streams, one wave per stream (AECM_PIPELINED=0)', round(d['value']/1e6,1), 'M frames/s;', d['roofline']['kernel'])"
python tools/bench_host_io.py 2>&1 | tail -1
python tools/bench_single_session.py | tail -1
python tools/bench_sessions.py --streams 65536 --ticks 300 | tail -1
python tools/bench_sessions.py --streams 65536 --ticks 300 --async | tail -1
python tools/bench_sessions.py --streams 65536 --ticks 300 --classes 400 | tail -1
python tools/bench_sessions.py --streams 65536 --ticks 200 --host | tail -1
python tools/bench_sessions.py --streams 65536 --ticks 200 --pinned | tail -1
python tools/bench_sessions.py --streams 65536 --fs 8000 --ticks 300 | tail -1
for s in 1024 8192; do python tools/bench_sessions.py --streams $s --ticks 300 | tail -1; done
} > $O/r4f_sweep.log 2>&1
tail -4 $O/r4f_pytest.log; tail -1 $O/r4f_bench.log | cut -c1-400; cat $O/r4f_sweep.log
