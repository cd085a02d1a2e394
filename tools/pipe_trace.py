#!/usr/bin/env python3
"""Run ON THE GPU BOX with a diagnostics build of the library (tools/ab_build.py trace="-DAECM_PIPE_TRACE ..."):
where the time of a pipelined launch goes, per wave -- when each workgroup started and finished (100 MHz wall clock) and how
long each of its six waves sat at the per-block barrier (shader clocks).

    AECM_LIB_PATH=webrtc_aecm_amd/_lib/ab_trace.so python tools/pipe_trace.py --streams 4096 --blocks 2048
"""
import argparse
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--blocks", type=int, default=2048)
    ap.add_argument("--fs", type=int, default=16000)
    ap.add_argument("--policy", default="", help="launch-policy fields to change: 'field=value ...' of AecmLaunchPolicy")
    a = ap.parse_args()
    trace = tempfile.NamedTemporaryFile(suffix=".bin", delete=False).name
    os.environ["AECM_PIPE_TRACE_FILE"] = trace
    import torch

    import bench
    import webrtc_aecm_amd as aecm
    S, T = a.streams, a.blocks
    dev = torch.device("cuda", 0)
    far, near = bench.synth_on_device(torch, S, T * 64, 1234, dev)
    out = torch.empty_like(near)
    b = aecm.AecmBatch(S, a.fs, cng_mode=1, echo_mode=1)
    if a.policy:
        b.set_launch_policy(**{kv.split("=", 1)[0]: int(kv.split("=", 1)[1], 0) for kv in a.policy.split()})
    assert b.describe_launch(T)[0] == 3, "not a pipelined launch"
    shape = b.describe_launch(T)[1]
    gain_waves = 4 if shape & 0x1000 else 0
    tail_waves, front_waves, delay_waves = shape & 0xff, 4 if shape & 0x200 else 2, ((2 if gain_waves else 4) if shape & 0x800 else 0)
    torch.cuda.synchronize()
    for _ in range(2):                       # the second launch (steady state of the signal's second pass) is the one recorded
        b.process_device(far.data_ptr(), near.data_ptr(), out.data_ptr(), T * 64, 64, T)
        b.synchronize()
    ms = b.last_launch_ms()
    rec = np.fromfile(trace, dtype=np.uint64).reshape(-1, 16, 4).astype(np.int64)       # [workgroup][wave][t0, t1, wait, total]; waves a workgroup does not have read 0
    n_wg = rec.shape[0]                      # one record set per workgroup of the launch (workgroups need not hold four streams)
    hw_id, xcc = rec[:, :, 2] >> 40, (rec[:, :, 3] >> 40) & 0xf           # placement (see the kernel's trace epilogue)
    rec[:, :, 2] &= (1 << 40) - 1
    rec[:, :, 3] &= (1 << 40) - 1
    n_waves = int((rec[0, :, 3] > 0).sum())
    rec = rec[:, :n_waves]
    t0 = rec[:, :, 0].min()
    start = (rec[:, :, 0].min(axis=1) - t0) / 100.0          # us
    end = (rec[:, :, 1].max(axis=1) - t0) / 100.0
    wait_frac = rec[:, :, 2] / np.maximum(rec[:, :, 3], 1)
    q = lambda x: [round(float(v), 1) for v in np.percentile(x, [0, 10, 50, 90, 100])]
    res = {
        "streams": S, "blocks": T, "tail_waves": tail_waves, "kernel_ms": ms, "workgroups": int(n_wg),
        "start_us_pctl_0_10_50_90_100": q(start), "end_us_pctl_0_10_50_90_100": q(end),
        "mean_end_over_last_end": round(float(end.mean() / end.max()), 4),
        "barrier_wait_fraction_back_waves_mean": round(float(wait_frac[:, :4].mean()), 4),
        "waves_per_workgroup": n_waves,
        "front_waves": front_waves, "delay_waves": delay_waves,
        "barrier_wait_fraction_front_waves_mean": round(float(wait_frac[:, 4:4 + front_waves].mean()), 4),
        "barrier_wait_fraction_tail_waves_mean": round(float(wait_frac[:, 4 + front_waves:4 + front_waves + tail_waves].mean()), 4) if tail_waves else None,
        "barrier_wait_fraction_delay_waves_mean": round(float(wait_frac[:, 4 + front_waves + tail_waves:4 + front_waves + tail_waves + delay_waves].mean()), 4) if delay_waves else None,
        "gain_waves": gain_waves,
        "barrier_wait_fraction_gain_waves_mean": round(float(wait_frac[:, 4 + front_waves + tail_waves + delay_waves:].mean()), 4) if gain_waves else None,
        "barrier_wait_fraction_back_pctl": [round(float(v), 3) for v in np.percentile(wait_frac[:, :4], [0, 10, 50, 90, 100])],
        "barrier_wait_fraction_front_pctl": [round(float(v), 3) for v in np.percentile(wait_frac[:, 4:4 + front_waves], [0, 10, 50, 90, 100])],
        # do the workgroups that finish early share something?  finish time by dispatch order (blockIdx) in eight bands
        "end_us_by_blockidx_octile": [round(float(v), 1) for v in end.reshape(-1)[: n_wg // 8 * 8].reshape(8, -1).mean(axis=1)] if n_wg >= 8 else None,
        "end_us_by_blockidx_mod8": [round(float(end[k::8].mean()), 1) for k in range(8)] if n_wg >= 8 else None,
    }
    # placement: per CU and SIMD, how many waves of each role (vector work per step differs by role: a SIMD with more back waves has more to do)
    n_back = 4
    role = np.zeros(n_waves, dtype=int)                  # 0 back / middle / channel, 1 front, 2 tail, 3 delay, 4 gain
    role[n_back:n_back + front_waves] = 1
    role[n_back + front_waves:n_back + front_waves + tail_waves] = 2
    role[n_back + front_waves + tail_waves:n_back + front_waves + tail_waves + delay_waves] = 3
    role[n_back + front_waves + tail_waves + delay_waves:] = 4
    hw, xc = hw_id[:, :n_waves], xcc[:, :n_waves]
    cu_key = (xc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 0xf)
    simd = (hw >> 4) & 3
    per = {}
    for w in range(n_wg):
        for v in range(n_waves):
            per.setdefault(int(cu_key[w, v]), np.zeros((4, 5), dtype=int))[int(simd[w, v]), role[v]] += 1
    mixes = {}
    for m in per.values():
        for sd in range(4):
            k = "/".join(str(int(x)) for x in m[sd])
            mixes[k] = mixes.get(k, 0) + 1
    res["cus_used"] = len(per)
    res["workgroup_waves_on_one_cu"] = bool(all(len(set(cu_key[w].tolist())) == 1 for w in range(n_wg)))
    res["simd_mixes_back/front/tail/delay/gain_count"] = dict(sorted(mixes.items(), key=lambda kv: -kv[1])[:16])
    res["first_workgroups_simd_of_each_wave"] = [simd[w].tolist() for w in range(min(n_wg, 6))]
    res["first_workgroups_cu"] = [hex(int(cu_key[w, 0])) for w in range(min(n_wg, 12))]
    # which workgroups share a CU (by dispatch index) and where their waves sit: the CUs of workgroups 0, 1 and 9
    res["cu_mates"] = {}
    for w0 in (0, 1, 9):
        if w0 < n_wg:
            mates = [w for w in range(n_wg) if cu_key[w, 0] == cu_key[w0, 0]]
            res["cu_mates"][str(w0)] = {str(w): simd[w].tolist() for w in mates}
    res["workgroups_per_cu_histogram"] = {str(k): int(v) for k, v in zip(*np.unique([sum(1 for w in range(n_wg) if cu_key[w, 0] == c) for c in per], return_counts=True))}
    print(json.dumps(res))
    os.unlink(trace)


if __name__ == "__main__":
    main()
