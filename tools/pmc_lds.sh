#!/bin/bash
# Run ON THE GPU BOX: LDS counters of the block kernel for A/B library variants (webrtc_aecm_amd/_lib/ab_<v>.so):
# which LDS access pattern produces the bank-conflict cycles, and does the LDS pipe ever hold the waves up?
#   tools/pmc_lds.sh variant1 variant2 ...      -> gpurun_out/pmc_lds/<variant>_{a,b}/...csv + a summary on stdout
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_lds
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export AECM_LIB_PATH=$R/webrtc_aecm_amd/_lib/ab_$v.so
  B="python $R/bench.py --no-cpu-baseline --no-parity --steps 2 --warmup 1"
  timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES \
      --kernel-trace --output-format csv -d "$OUT/${v}_a" -o bench -- $B > "$OUT/${v}_a.log" 2>&1
  timeout 600 rocprofv3 --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d "$OUT/${v}_b" -o bench -- $B > "$OUT/${v}_b.log" 2>&1
done
python3 - "$OUT" "$@" <<'PY'
import collections, csv, sys
from pathlib import Path
out = Path(sys.argv[1])
for v in sys.argv[2:]:
    agg = collections.defaultdict(list)
    for part in ("a", "b"):
        for f in (out / f"{v}_{part}").rglob("*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "aecm_process" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    frames = 65536 * 1280
    print(v, {k: round(sum(x) / len(x) / frames, 3) for k, x in sorted(agg.items())})
PY
