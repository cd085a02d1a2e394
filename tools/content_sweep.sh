#!/bin/bash
# Run ON THE GPU BOX: the headline (and BASELINE configs[1], and a 1 024-stream launch) on each content profile of bench.py --profile, every run with its own
# 18-stream parity check against the CPU checker (the unmodified reference when oracle/_ref travelled) and the shares of the
# blocks that took the data-dependent paths (counted by the restatement's branch counters over the timed passes).
#   gpurun -- 'bash tools/content_sweep.sh > gpurun_out/content_sweep.txt'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
for args in "" "--streams 4096 --blocks 2048" ${CONTENT_SWEEP_EXTRA:+"$CONTENT_SWEEP_EXTRA"} "--streams 1024 --blocks 2048"; do
  for p in recipe always_active double_talk full_scale silent; do
    timeout 600 python bench.py --no-cpu-baseline --profile $p $args 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c, p = d['content'], d['parity']
print(f\"{d['config']['streams_per_gpu']:>6} streams  {c['profile']:<14} {d['value'] / 1e6:8.1f} M frames/s  {d['ms_per_step']:8.3f} ms/step  \"
      f\"nlms {c['nlms_share']:.3f}  passthrough {c['passthrough_share']:.3f}  q_steady {c['q_steady_share']:.3f}  ifft_unscaled {c['ifft_unscaled_share']:.3f}  \"
      f\"delayed {c['delayed_share']:.3f}  parity {'ok' if p['ok'] else 'FAILED'} ({p['checker']}, {len(p['streams'])} streams x {p['blocks']} blocks)  {d['roofline']['kernel']}\")
"
  done
done
