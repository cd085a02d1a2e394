#!/bin/bash
# Run ON THE GPU BOX: interleaved A/B timing of library builds and / or launch policies -- the one script behind every
# "x -> y M frames/s" of profiles/r0N_experiments.md.
#
#   tools/ab.sh [variant ...]
#       variant = name of an A/B build (webrtc_aecm_amd/_lib/ab_<name>.so, made by tools/ab_build.py), "shipped" for the product
#                 library, or a path to a .so
#   environment:
#     BENCH_ARGS   arguments for bench.py (default: the headline workload), e.g. "--streams 4096 --blocks 2048"
#     POLICIES     launch policies to cross with the variants, separated by ';' -- fields of AecmLaunchPolicy (include/aecm_batch.h),
#                  e.g. POLICIES="; queue_chunk_blocks=0; pipelined_min_streams=0" (the empty entry = the shipped policy)
#     REPS / STEPS repetitions (3) and timed launches per run (8)
#     PMC          a rocprofv3 counter list: one counter pass per variant instead of the timing loop (SQ_* per launch of the dominant kernel)
#     SESSIONS     arguments for tools/bench_sessions.py: time that instead of bench.py (the tick kernels)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
[ $# -eq 0 ] && set -- shipped
lib_of() { case "$1" in shipped) echo webrtc_aecm_amd/_lib/libaecm_mi355x.so;; */*|*.so) echo "$1";; *) echo webrtc_aecm_amd/_lib/ab_$1.so;; esac; }
IFS=';' read -r -a POL <<< "${POLICIES:-}"
[ ${#POL[@]} -eq 0 ] && POL=("")
if [ -n "${PMC:-}" ]; then
  OUT=$R/gpurun_out; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
  for v in "$@"; do
    tag=$(basename "$(lib_of $v)" .so)
    AECM_LIB_PATH=$R/$(lib_of $v) rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc_$tag" -o bench -- \
        python $R/bench.py --no-cpu-baseline --no-parity --steps 3 --warmup 1 ${BENCH_ARGS:-} > "$OUT/pmc_$tag.log" 2>&1
    python - "$OUT/pmc_$tag" "$v" <<'PY'
import collections, csv, glob, sys
acc, dur = collections.defaultdict(lambda: collections.defaultdict(list)), collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "aecm_" in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    print(sys.argv[2], k, {n: sum(x) / len(x) for n, x in sorted(c.items())})
PY
  done
  exit 0
fi
for rep in $(seq 1 ${REPS:-3}); do
  for v in "$@"; do
    for p in "${POL[@]}"; do
      p=$(echo $p)        # trim
      if [ -n "${SESSIONS:-}" ]; then
        AECM_LIB_PATH=$(lib_of $v) timeout 300 python tools/bench_sessions.py $SESSIONS 2>&1 | tail -1 |
          python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_tick'],4), 'ms per tick', d['realtime_streams_per_gpu'], 'real-time streams')"
      else
        AECM_LIB_PATH=$(lib_of $v) timeout 300 python bench.py --no-cpu-baseline --no-parity --steps ${STEPS:-8} --warmup 2 ${BENCH_ARGS:-} ${p:+--policy "$p"} 2>&1 | tail -1 |
          python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '[$p]', round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step'],3), 'ms', d['roofline']['kernel'])"
      fi
    done
  done
done
