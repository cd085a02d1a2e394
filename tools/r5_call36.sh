#!/bin/bash
# Run ON THE GPU BOX (round 5, call 36): every frame of small launches (the sixteen-wave shape) against the reference: 8 kHz, odd batch sizes, cng off, other echo modes, long launches.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
O=gpurun_out
mkdir -p $O
{
python tools/soak_parity.py --streams 1024 --blocks 2048 --fs 8000 2>&1 | tail -1
python tools/soak_parity.py --streams 1001 --blocks 1500 --passes 3 --seed 77 2>&1 | tail -1
python tools/soak_parity.py --streams 37 --blocks 4096 --cng 0 --echo-mode 0 --seed 5 2>&1 | tail -1
python tools/soak_parity.py --streams 514 --blocks 2048 --echo-mode 4 --seed 9 2>&1 | tail -1
python tools/soak_parity.py --streams 1536 --blocks 2048 --seed 11 2>&1 | tail -1
} > $O/r5_call36.jsonl 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r5_call36.jsonl'):
    if l.startswith('{'):
        d = json.loads(l); print(d['streams'], d['fs'], 'cng', d['cng'], 'echo', d['echo_mode'], d['checker'], 'ok', d['ok'], 'samples', d['samples_compared'], 'digests', d['digests_compared'], d['launch_form'])
    else:
        print(l.strip()[:200])
PY
