#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
for m in flow lean; do
  AECM_TICK_MODE=$m rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pm_$m -o t -- python tools/bench_sessions.py --streams 65536 --ticks 60 > gpurun_out/pm_$m.log 2>&1
  python - <<PY
import csv,collections,glob
f=glob.glob('gpurun_out/pm_$m/**/t_counter_collection.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'aecm_tick' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
print('$m', {k: round(sum(v[-40:])/40/65536,1) for k,v in d.items()}, 'per session-tick')
PY
done
