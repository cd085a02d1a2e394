#!/bin/bash
# Run ON THE GPU BOX: SQ counter passes of bench.py for A/B library variants.
#   tools/pmc_ab.sh "<counter list>" tag variant1 variant2 ...      (variants: webrtc_aecm_amd/_lib/ab_<v>.so)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p "$OUT"
COUNTERS=$1; TAG=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  AECM_LIB_PATH=$R/webrtc_aecm_amd/_lib/ab_$v.so rocprofv3 --pmc $COUNTERS --kernel-trace --output-format csv \
      -d "$OUT/pmc_${TAG}_$v" -o bench -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/pmc_${TAG}_$v.log" 2>&1
  python - "$OUT/pmc_${TAG}_$v" "$v" <<'PY'
import csv, sys, glob, collections
d, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "aecm_process_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(v, {k: sum(x) / len(x) for k, x in sorted(acc.items())})
PY
done
