#!/bin/bash
# Run ON A GPU BOX with fewer GPUs than the scaling run will use: the exact command shape of the driver's N = 2 / 4 / 8 runs
# (python bench.py --gpus N re-executes itself under torch.distributed.run), with the ranks mapped onto the devices present
# (--share-devices: gloo for the counter collectives, RCCL refuses two ranks on one device).  Checks per N: every rank took part
# (ranks_seen), every rank owns its own shard, every rank's timed workload is bit-exact against the CPU checker.
#   gpurun -- 'bash tools/scale_preflight.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
rc=0
for n in ${GPUS:-2 4 8}; do
  line=$(timeout 900 python bench.py --gpus $n --share-devices --streams ${STREAMS:-2048} --blocks ${BLOCKS:-128} --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{' | tail -1)
  python - "$n" "$line" <<'PY' || rc=1
import json, sys
n, d = int(sys.argv[1]), json.loads(sys.argv[2])
r, p = d["ranks"], d["parity"]
ok = d["n_gpus"] == n and r["ranks_seen"] == n and len(r["per_rank_streams"]) == n and p["ok"] and p["ranks_ok"] == n
print(f"N={n}: ranks_seen {r['ranks_seen']}, per-rank streams {r['per_rank_streams']}, parity ranks_ok {p['ranks_ok']}/{n} "
      f"({p['checker']}, {p['checker_threads']} checker threads per rank), {d['value'] / 1e6:.1f} M frames/s aggregate on the shared device -> {'OK' if ok else 'FAILED'}")
sys.exit(0 if ok else 1)
PY
done
exit $rc
