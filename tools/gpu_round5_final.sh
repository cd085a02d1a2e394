#!/bin/bash
# Run ON THE GPU BOX: the round's closing measurements (tests, bench line, profiles of the headline and of the configs[1] kernel,
# content sweep, config sweep, serving-path numbers).
#   gpurun --timeout 2400 -- 'bash tools/gpu_round5_final.sh'
#   then here: python tools/summarize_profile.py r05 && python tools/summarize_profile.py r05_pipelined ppipe && python tools/summarize_profile.py r05_small psmall
set -u
O=gpurun_out
mkdir -p $O
( time AECM_SANITIZER_LOG=$PWD/$O/r5f_ubsan_gpu.log python -m pytest tests -m gpu -x -q --durations=6 ) > $O/r5f_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r5f_pytest.log
python bench.py --steps 20 --warmup 5 > $O/r5f_bench.log 2>&1
PARTS="stats hbm sq cal tick" bash tools/profile_gpu.sh > $O/r5f_profile.log 2>&1
PREFIX=ppipe PARTS="stats hbm sq cal" BENCH_ARGS="--streams 4096 --blocks 2048" CENSUS_KERNEL=aecm_process_pipelined_kernelILi0ELb1ELb1ELi2E \
  bash tools/profile_gpu.sh > $O/r5f_profile_pipelined.log 2>&1
PREFIX=psmall PARTS="stats hbm sq cal" BENCH_ARGS="--streams 1024 --blocks 2048" CENSUS_KERNEL=aecm_process_pipelined_kernelILi2ELb0ELb0ELi4ELi2ELi4E \
  bash tools/profile_gpu.sh > $O/r5f_profile_small.log 2>&1
bash tools/content_sweep.sh > $O/r5f_content_sweep.txt 2>&1
{
for a in "--streams 4 --blocks 2048" "--streams 64 --blocks 2048" "--streams 256 --blocks 2048" "--streams 512 --blocks 2048" "--streams 1024 --blocks 2048" "--streams 1536 --blocks 2048" "--streams 2048 --blocks 2048" "--streams 3072 --blocks 2048" \
         "--streams 4096 --blocks 2048" "--streams 6144" "--streams 16384" "--fs 8000 --streams 32768" "--streams 131072 --blocks 512" "--clean" "--variant safe"; do
  python bench.py --no-cpu-baseline $a | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', round(d['value']/1e6,1), 'M frames/s; parity', d['parity']['ok'], ';', d['roofline']['kernel'], ';', d['config']['workload'][:70])"
done
AECM_QUEUE_CHUNK=0 python bench.py --no-cpu-baseline --no-parity | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one wave per stream (AECM_QUEUE_CHUNK=0)', round(d['value']/1e6,1), 'M frames/s;', d['roofline']['kernel'])"
AECM_PIPELINED=0 python bench.py --no-cpu-baseline --no-parity --streams 4096 --blocks 2048 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4096 streams, one wave per stream (AECM_PIPELINED=0)', round(d['value']/1e6,1), 'M frames/s;', d['roofline']['kernel'])"
python tools/bench_host_io.py 2>&1 | tail -1
python tools/bench_single_session.py | tail -1
python tools/bench_sessions.py --streams 65536 --ticks 300 | tail -1
python tools/bench_sessions.py --streams 65536 --ticks 300 --async | tail -1
python tools/bench_sessions.py --streams 65536 --fs 8000 --ticks 300 | tail -1
for s in 1024 8192; do python tools/bench_sessions.py --streams $s --ticks 300 | tail -1; done
( python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "65536_streams_is_fast" 2>&1 | grep "65 536 states" )
bash tools/scale_preflight.sh
} > $O/r5f_sweep.log 2>&1
for s in 64 1024 2048 3072 4096; do python tools/soak_parity.py --streams $s --blocks 2048 2>&1 | tail -1; done > $O/r5f_soak_pipelined.jsonl 2>&1
tail -4 $O/r5f_pytest.log; tail -1 $O/r5f_bench.log | cut -c1-400; cat $O/r5f_content_sweep.txt $O/r5f_sweep.log; cat $O/r5f_soak_pipelined.jsonl | cut -c1-300
