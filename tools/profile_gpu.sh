#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py (and of the streaming
# tick bench).  Outputs land in gpurun_out/prof_*; tools/summarize_profile.py condenses them into profiles/.
#   PARTS="stats hbm sq cal tick pcs" tools/profile_gpu.sh      (default: stats hbm sq cal tick)
# Counters are collected in their own runs with --kernel-trace only (never together with sys/hip/hsa tracing).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PARTS=${PARTS:-"stats hbm sq cal tick"}
P=${PREFIX:-prof}       # output directories gpurun_out/${P}_* (a second profile of another workload in the same call: PREFIX=ppipe)
BENCH="python $R/bench.py --no-cpu-baseline --no-parity ${BENCH_ARGS:-}"
PMC_STEPS="--steps 2 --warmup 1"
has() { [[ " $PARTS " == *" $1 "* ]]; }
pmc() {   # pmc <dir> <counters...>
  local d=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$d" -o bench -- $BENCH $PMC_STEPS > "$OUT/$d.log" 2>&1
}
python - > "$OUT/${P}_meta.json" <<PY
import json, subprocess, sys
sys.path.insert(0, "$R")
import webrtc_aecm_amd as aecm
from webrtc_aecm_amd import isa_census
lib = aecm.load()
c = isa_census.census(aecm.library_path(), "${CENSUS_KERNEL:-}" or isa_census.HEADLINE_KERNEL)
from webrtc_aecm_amd import build as _b
bi = _b.build_info()
commit = (bi.get("commit") or "unknown") + ("+dirty" if bi.get("dirty") else "")
print(json.dumps({"state_size_bytes": lib.WebRtcAecmBatch_state_size_bytes(), "kernel_symbol": c["kernel"], "kernel_fingerprint": c["fingerprint"],
                  "static_counts": c["counts"], "static_valu_fast_class": c["valu_fast_class"], "commit": commit or None}))
PY
if has stats; then   # per-kernel time (same command as the bench line)
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${P}_stats" -o bench -- $BENCH --steps 10 --warmup 2 > "$OUT/${P}_stats.log" 2>&1
fi
if has hbm; then     # HBM traffic: separate passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
  pmc ${P}_fetch FETCH_SIZE
  pmc ${P}_write WRITE_SIZE
fi
if has sq; then      # SQ counters: instruction mix, busy cycles, LDS bank conflicts
  pmc ${P}_sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  pmc ${P}_sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  pmc ${P}_sq3 SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_INT32 SQ_IFETCH
  pmc ${P}_grbm GRBM_GUI_ACTIVE GRBM_COUNT
fi
if has cal; then     # FETCH_SIZE calibration on a known byte count: fixed delay 0 => no far-history reads, so the
                     # kernel reads exactly inputs (256 B/frame) + one state image per stream per launch
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${P}_fetch_cal" -o bench -- $BENCH $PMC_STEPS --fixed-delay 0 > "$OUT/${P}_fetch_cal.log" 2>&1
fi
if has tick; then    # the streaming path: 65 536 sessions on a 10 ms clock
  TICK="python $R/tools/bench_sessions.py --streams 65536 --ticks 200"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${P}_tick" -o tick -- $TICK > "$OUT/${P}_tick.log" 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${P}_tick_fetch" -o tick -- $TICK > "$OUT/${P}_tick_fetch.log" 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/${P}_tick_write" -o tick -- $TICK > "$OUT/${P}_tick_write.log" 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/${P}_tick_sq" -o tick -- $TICK > "$OUT/${P}_tick_sq.log" 2>&1
fi
if has pcs; then     # PC sampling of the block kernel (beta feature; bounded by a short timeout of its own)
  ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 180 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap \
      --pc-sampling-interval 1000 --kernel-trace --output-format csv -d "$OUT/${P}_pcs" -o bench -- $BENCH --steps 3 --warmup 1 --streams 16384 --blocks 256 > "$OUT/${P}_pcs.log" 2>&1
  echo "pc sampling rc=$?" >> "$OUT/${P}_pcs.log"
fi
find "$OUT" -name "*.csv" -newer "$OUT/${P}_meta.json" | head -60
