set -u
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r2a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a_pytest.log
python bench.py > gpurun_out/r2a_bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/r2a_bench.log
tools/valu_rate_microbench --batch3 > gpurun_out/r2a_microbench3.txt 2>&1
PARTS="stats hbm sq cal tick" tools/profile_gpu.sh > gpurun_out/r2a_profile.log 2>&1
python tools/bench_sessions.py --streams 65536 --ticks 300 > gpurun_out/r2a_sessions.log 2>&1
tail -3 gpurun_out/r2a_pytest.log; tail -1 gpurun_out/r2a_bench.log | cut -c1-600; cat gpurun_out/r2a_microbench3.txt; tail -1 gpurun_out/r2a_sessions.log
