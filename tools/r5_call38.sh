#!/bin/bash
# Run ON THE GPU BOX (round 5, call 38): rocprofv3 records of the mid-size shapes' kernels (2 048 and 3 072 streams).
#   then here: python tools/summarize_profile.py r05_mid2048 pmid2k && python tools/summarize_profile.py r05_mid3072 pmid3k
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
PREFIX=pmid2k PARTS="stats hbm sq cal" BENCH_ARGS="--streams 2048 --blocks 2048" CENSUS_KERNEL=aecm_process_pipelined_kernelILi2ELb0ELb1ELi4ELi0ELi0E \
  bash tools/profile_gpu.sh > $O/r5_call38_2k.log 2>&1
PREFIX=pmid3k PARTS="stats hbm sq cal" BENCH_ARGS="--streams 3072 --blocks 2048" CENSUS_KERNEL=aecm_process_pipelined_kernelILi2ELb0ELb1ELi2ELi0ELi0E \
  bash tools/profile_gpu.sh > $O/r5_call38_3k.log 2>&1
tail -2 $O/r5_call38_2k.log $O/r5_call38_3k.log
