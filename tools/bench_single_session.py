#!/usr/bin/env python3
"""Latency of the drop-in single-session ABI (WebRtcAecm_BufferFarend + WebRtcAecm_Process per 10 ms call: H2D,
one launch, D2H, synchronise) next to the reference on one host core, and the session count at which one GPU tick of
S concurrent sessions (WebRtcAecmSessions_Tick) becomes cheaper per session than one host core.  A row for DESIGN.md /
INTEGRATION.md, not the headline metric."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def time_calls(sess, far, near, frame, n_calls, ms=40):
    t = []
    for i in range(n_calls):
        sl = slice(i * frame, (i + 1) * frame)
        t0 = time.perf_counter()
        sess.buffer_farend(far[sl])
        sess.process(near[sl], None, ms)
        t.append(time.perf_counter() - t0)
    t = np.array(t[n_calls // 5:])                       # past the start-up phase (pass-through calls cost nothing)
    return float(np.median(t) * 1e6), float(np.percentile(t, 99) * 1e6)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fs", type=int, default=16000)
    ap.add_argument("--calls", type=int, default=1500)
    args = ap.parse_args()
    import torch

    import webrtc_aecm_amd as aecm
    from oracle import pyoracle
    from webrtc_aecm_amd.synth import synth_pair
    fs, frame = args.fs, args.fs // 100
    far, near = synth_pair(3, args.calls * frame // 64 + 1, fs, "steady")
    res = {"fs": fs, "samples_per_call": frame}
    s = aecm.Aecm()
    assert s.init(fs) == 0 and s.set_config(1, 1) == 0
    res["gpu_single_session_us_per_call_median_p99"] = time_calls(s, far, near, frame, args.calls)
    s.close()
    if pyoracle.have_reference():
        r = pyoracle.RefSession(fs, 1, 1)
        res["reference_one_core_us_per_call_median_p99"] = time_calls(r, far, near, frame, args.calls)
    # batched sessions: wall time of one tick of S sessions, device-resident audio
    ticks = {}
    for S in (1, 16, 64, 256, 1024, 4096, 16384, 65536):
        g = torch.Generator(device="cuda").manual_seed(1)
        f = (torch.randn((S, frame * 8), generator=g, device="cuda") * 3000).clamp_(-32768, 32767).to(torch.int16)
        d = (f.roll(37, dims=1) // 3 + (torch.randn((S, frame * 8), generator=g, device="cuda") * 200).to(torch.int16))
        o = torch.empty_like(f)
        sb = aecm.AecmSessions(S, fs, 1, 1)
        torch.cuda.synchronize()
        n = 200 if S <= 4096 else 100
        for i in range(40 + n):
            if i == 40:
                t0 = time.perf_counter()
            off = (i % 8) * frame * 2
            assert sb.tick_device(f.data_ptr() + off, d.data_ptr() + off, o.data_ptr(), f.shape[1], frame, 40) == 0
        ticks[S] = (time.perf_counter() - t0) / n * 1e6
        sb.close()
    res["gpu_tick_us_by_sessions"] = ticks
    if "reference_one_core_us_per_call_median_p99" in res:
        cpu = res["reference_one_core_us_per_call_median_p99"][0]
        res["sessions_per_tick_where_gpu_beats_one_core"] = next((S for S, us in ticks.items() if us / S < cpu), None)
        res["sessions_one_core_serves_in_10ms"] = int(10000 / cpu)
        res["sessions_one_gpu_serves_in_10ms"] = int(65536 * 10000 / ticks[65536])
    print(json.dumps(res))


if __name__ == "__main__":
    main()
