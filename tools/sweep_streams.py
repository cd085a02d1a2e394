#!/usr/bin/env python3
"""Run ON THE GPU BOX: frames/s of WebRtcAecmBatch_ProcessBlocks over a range of batch sizes, in ONE process (no start-up
cost per point), optionally for several launch policies side by side.

  python tools/sweep_streams.py --sizes 256:4096:256 --blocks 2048                      # the shipped launch policy
  python tools/sweep_streams.py --sizes 1280,1536 --set pipe_spread=0 --set "pipe_gain_waves=0 pipe_delay_waves=0 pipe_front_waves=4 pipe_raw=1"
  python tools/sweep_streams.py --sizes 256:8192:256 --check-monotone                    # exit 1 if frames/s ever falls as S grows

Each --set is one column: field=value pairs (space separated) of AecmLaunchPolicy (include/aecm_batch.h) set on the batch's
policy through WebRtcAecmBatch_SetLaunchPolicy -- no environment variable, no special build.
Every point: `--reps` repetitions of `--steps` timed launches after 2 warm-up launches; the best repetition is printed (kernel
time from the library's own HIP events).  Output: one line per size, and a JSON record per point on stderr with --json.
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def parse_sizes(text):
    out = []
    for part in text.split(","):
        if ":" in part:
            a, b, c = (int(x) for x in part.split(":"))
            out += list(range(a, b + 1, c))
        else:
            out.append(int(part))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="256:4096:256")
    ap.add_argument("--blocks", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--fs", type=int, default=16000)
    ap.add_argument("--profile", default="recipe")
    ap.add_argument("--set", action="append", default=[], help="one column: 'field=value field2=value' of AecmLaunchPolicy (default: one column, the shipped policy)")
    ap.add_argument("--check-monotone", action="store_true", help="exit 1 when a column's frames/s falls by more than --tolerance as S grows")
    ap.add_argument("--tolerance", type=float, default=0.01)
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()

    import torch

    import webrtc_aecm_amd as aecm
    from bench import synth_on_device

    sizes = parse_sizes(args.sizes)
    columns = args.set or [""]
    device = torch.device("cuda", 0)
    T = args.blocks
    far_all, near_all = synth_on_device(torch, max(sizes), T * 64, 1234, device, profile=args.profile)
    out_all = torch.empty_like(near_all)
    stride = far_all.shape[1]
    results = {c: [] for c in columns}
    print("streams  " + "  ".join(f"[{c or 'shipped'}]" for c in columns))
    for S in sizes:
        cells = []
        for col in columns:
            batch = aecm.AecmBatch(S, args.fs, cng_mode=1, echo_mode=1, device=0)
            wishes = {kv.split("=", 1)[0]: int(kv.split("=", 1)[1], 0) for kv in col.split()}
            if wishes:
                batch.set_launch_policy(**wishes)
            form, shape = batch.describe_launch(T, False)
            best = 0.0
            for _ in range(args.reps):
                for _ in range(2):
                    batch.process_device(far_all.data_ptr(), near_all.data_ptr(), out_all.data_ptr(), stride, 64, T)
                batch.synchronize()
                batch.reset_timers()
                for _ in range(args.steps):
                    batch.process_device(far_all.data_ptr(), near_all.data_ptr(), out_all.data_ptr(), stride, 64, T)
                batch.synchronize()
                ms, launches = batch.timers()
                best = max(best, S * T * launches / (ms / 1e3))
            batch.close()
            results[col].append(best)
            cells.append(f"{best / 1e6:8.1f} M (form {form}, shape {shape:#x})")
            if args.json:
                print(json.dumps({"streams": S, "blocks": T, "fs": args.fs, "set": col, "frames_per_s": best, "form": form, "shape": shape}), file=sys.stderr)
        print(f"{S:7d}  " + "  ".join(cells), flush=True)
    bad = []
    for col in columns:
        r = results[col]
        for i in range(1, len(r)):
            if r[i] < max(r[:i]) * (1.0 - args.tolerance):
                bad.append((col or "shipped", sizes[i], r[i], max(r[:i])))
    for col, S, v, before in bad:
        print(f"NOT MONOTONE [{col}]: {S} streams {v / 1e6:.1f} M < {before / 1e6:.1f} M at a smaller size")
    if not bad:
        print("monotone: frames/s never falls as the batch grows" + (f" (tolerance {args.tolerance:.0%})" if args.tolerance else ""))
    if args.check_monotone and bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
