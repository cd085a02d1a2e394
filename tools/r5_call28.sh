#!/bin/bash
# Run ON THE GPU BOX (round 5, call 28): content profiles on the small launches' sixteen-wave shape (each with its parity check), 8 kHz.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
{
for p in recipe always_active double_talk full_scale silent; do
  timeout 600 python bench.py --no-cpu-baseline --profile $p --streams 1024 --blocks 2048 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c, p = d['content'], d['parity']
print(f\"{d['config']['streams_per_gpu']:>6} streams  {c['profile']:<14} {d['value'] / 1e6:8.1f} M frames/s  {d['ms_per_step']:8.3f} ms/step  \"
      f\"nlms {c['nlms_share']:.3f}  passthrough {c['passthrough_share']:.3f}  q_steady {c['q_steady_share']:.3f}  ifft_unscaled {c['ifft_unscaled_share']:.3f}  \"
      f\"delayed {c['delayed_share']:.3f}  parity {'ok' if p['ok'] else 'FAILED'} ({p['checker']}, {len(p['streams'])} streams x {p['blocks']} blocks)  {d['roofline']['kernel']}\")
"
done
python bench.py --no-cpu-baseline --fs 8000 --streams 1024 --blocks 2048 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8 kHz 1024 streams', round(d['value']/1e6,1), 'M frames/s; parity', d['parity']['ok'], ';', d['roofline']['kernel'], '; issue_bound available', d['roofline']['issue_bound'].get('available'))"
python bench.py --no-cpu-baseline --streams 1024 --blocks 2048 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1024 streams', round(d['value']/1e6,1), 'M frames/s; parity', d['parity']['ok'], ';', d['roofline']['kernel'], '; issue_bound', json.dumps(d['roofline']['issue_bound'])[:600])"
python bench.py --steps 20 --warmup 5 | tail -1 > $O/r5_call28_bench_line.json
python -c "import json; d=json.load(open('$O/r5_call28_bench_line.json')); print('default line', round(d['value']/1e6,1), d['parity']['ok'], 'issue_bound available', d['roofline']['issue_bound'].get('available'), 'traffic', d['roofline']['traffic'], d['config']['commit'])"
} > $O/r5_call28.log 2>&1
cat $O/r5_call28.log
