#!/bin/bash
# Run ON THE GPU BOX: a round's closing measurements, everything the files under profiles/ are made from -- GPU tests, the bench
# line, rocprofv3 profiles of the headline kernel and of the pipelined kernels (4 096 / 1 024 streams), content sweep, batch-size
# sweep (with the monotonicity verdict), serving-path numbers (device-resident and host-fed), full-size soak parity, the multi-rank
# preflight.
#   gpurun --timeout 2700 -- 'TAG=r06 bash tools/gpu_round_final.sh'
#   then here: python tools/summarize_profile.py $TAG && python tools/summarize_profile.py ${TAG}_pipelined ppipe && python tools/summarize_profile.py ${TAG}_small psmall
set -u
T=${TAG:-rNN}
O=gpurun_out
mkdir -p $O
( time AECM_SANITIZER_LOG=$PWD/$O/${T}_ubsan_gpu.log python -m pytest tests -m gpu -x -q --durations=6 ) > $O/${T}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${T}_pytest.log
python bench.py --steps 20 --warmup 5 > $O/${T}_bench.log 2>&1
PARTS="stats hbm sq cal tick" bash tools/profile_gpu.sh > $O/${T}_profile.log 2>&1
kernel_of() { python - "$@" <<'PY'
import sys
sys.path.insert(0, ".")
import webrtc_aecm_amd as aecm
from webrtc_aecm_amd import isa_census
d = aecm.describe_launch_detail(int(sys.argv[1]), 2048, aecm.device_info(0)[1])
print(isa_census.block_kernel(d["form"], False, d["chunk_blocks"] if d["form"] == 2 else d["shape"])[0])
PY
}
PREFIX=ppipe PARTS="stats hbm sq cal" BENCH_ARGS="--streams 4096 --blocks 2048" CENSUS_KERNEL=$(kernel_of 4096) bash tools/profile_gpu.sh > $O/${T}_profile_pipelined.log 2>&1
PREFIX=psmall PARTS="stats hbm sq cal" BENCH_ARGS="--streams 1024 --blocks 2048" CENSUS_KERNEL=$(kernel_of 1024) bash tools/profile_gpu.sh > $O/${T}_profile_small.log 2>&1
bash tools/content_sweep.sh > $O/${T}_content_sweep.txt 2>&1
{
python tools/sweep_streams.py --sizes 256:8192:256 --blocks 2048 --tolerance 0.01 2>&1 | grep -v amdgpu.ids
echo "# between the multiples of the CU count (the static placement's sawtooth) and right behind each shape's largest launch:"
python tools/sweep_streams.py --sizes 1025,1100,1152,2049,2100,2176,3073,3150,3200,4097,4200 --blocks 2048 --tolerance 1.0 2>&1 | grep -v "amdgpu.ids\|monotone"
for a in "--streams 4 --blocks 2048" "--streams 64 --blocks 2048" "--streams 1024 --blocks 2048" "--streams 1536 --blocks 2048" "--streams 2560 --blocks 2048" \
         "--streams 4096 --blocks 2048" "--streams 16384" "--fs 8000 --streams 32768" "--streams 131072 --blocks 512" "--clean" "--variant safe"; do
  python bench.py --no-cpu-baseline $a | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', round(d['value']/1e6,1), 'M frames/s; parity', d['parity']['ok'], ';', d['roofline']['kernel'], ';', d['config']['workload'][:70])"
done
python bench.py --no-cpu-baseline --no-parity --policy "queue_chunk_blocks=0" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one wave per stream (policy queue_chunk_blocks=0)', round(d['value']/1e6,1), 'M frames/s;', d['roofline']['kernel'])"
python bench.py --no-cpu-baseline --no-parity --streams 4096 --blocks 2048 --policy "pipelined_min_streams=0" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4096 streams, one wave per stream (policy pipelined_min_streams=0)', round(d['value']/1e6,1), 'M frames/s;', d['roofline']['kernel'])"
python tools/bench_host_io.py 2>&1 | tail -1
python tools/bench_single_session.py | tail -1
( python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "65536_streams_is_fast" 2>&1 | grep "65 536 states" )
bash tools/scale_preflight.sh
} > $O/${T}_sweep.log 2>&1
{
for a in "--audio device" "--audio device --async" "--audio host" "--audio host-registered --async" "--audio host-staged"; do
  for s in 16384 65536; do python tools/bench_sessions.py --streams $s --ticks 300 $a 2>&1 | tail -1; done
done
python tools/bench_sessions.py --streams 65536 --fs 8000 --ticks 300 2>&1 | tail -1
python tools/bench_sessions.py --streams 65536 --fs 8000 --ticks 300 --audio host-staged 2>&1 | tail -1
for s in 1024 8192; do python tools/bench_sessions.py --streams $s --ticks 300 | tail -1; done
} > $O/${T}_sessions.txt 2>&1
for s in 64 768 1280 1536 2560 3584 4096; do python tools/soak_parity.py --streams $s --blocks 2048 2>&1 | tail -1; done > $O/${T}_soak_pipelined.jsonl 2>&1
python tools/soak_parity.py --streams 65536 --blocks 1280 --passes 2 2>&1 | tail -1 > $O/${T}_soak_headline.jsonl
tail -4 $O/${T}_pytest.log; tail -1 $O/${T}_bench.log | cut -c1-400; cat $O/${T}_content_sweep.txt $O/${T}_sweep.log $O/${T}_sessions.txt; cut -c1-300 $O/${T}_soak_pipelined.jsonl $O/${T}_soak_headline.jsonl
