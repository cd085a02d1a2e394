#!/usr/bin/env python3
"""Serving-shaped measurement: S concurrent WebRtcAecm_* sessions on a 10 ms clock
(WebRtcAecmSessions_Tick, device-resident audio).  Reports the time per tick and how many real-time
streams one GPU sustains.  Not the headline metric (bench.py is); a row for DESIGN.md."""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=65536)
    ap.add_argument("--fs", type=int, default=16000)
    ap.add_argument("--ticks", type=int, default=300)
    ap.add_argument("--classes", type=int, default=1,
                    help="distinct msInSndCardBuf values among the sessions (> 1: WebRtcAecmSessions_TickPerSession, one value per session)")
    ap.add_argument("--host", action="store_true", help="audio in host memory (WebRtcAecmSessions_TickHost): the PCIe-inclusive tick")
    ap.add_argument("--pinned", action="store_true",
                    help="audio in caller-owned host memory registered once (WebRtcAecmBatch_RegisterHostBuffer): the kernels read and "
                         "write it in place over the link, no staging copies")
    ap.add_argument("--async", dest="asynchronous", action="store_true",
                    help="WebRtcAecmSessions_TickAsync: ticks are enqueued back to back, one synchronisation at the end")
    args = ap.parse_args()
    import torch

    import webrtc_aecm_amd as aecm
    S, fs = args.streams, args.fs
    n = fs // 100                                     # one 10 ms tick
    g = torch.Generator(device="cuda").manual_seed(1)
    far = (torch.randn((S, n * 8), generator=g, device="cuda") * 3000).clamp_(-32768, 32767).to(torch.int16)
    near = (far.roll(37, dims=1) // 3 + (torch.randn((S, n * 8), generator=g, device="cuda") * 200).to(torch.int16))
    out = torch.empty_like(far)                       # Tick() uses ONE row stride for far, near and out
    sess = aecm.AecmSessions(S, fs, 1, 1)
    torch.cuda.synchronize()

    import numpy as np
    ms = (40 + (np.arange(S) % args.classes)).astype(np.int16)          # distinct values in [40, 40 + classes)

    if args.host or args.pinned:
        hfar = np.ascontiguousarray(far.cpu().numpy()[:, :n])
        hnear = np.ascontiguousarray(near.cpu().numpy()[:, :n])
    if args.pinned:
        hout = np.zeros_like(hnear)
        pf, pn, po = (aecm.register_host_buffer(a) for a in (hfar, hnear, hout))

    def tick(i):
        if args.pinned:
            rc = sess.tick_async(pf, pn, po, n, n, 40) if args.asynchronous else sess.tick_device(pf, pn, po, n, n, 40)
            assert rc == 0, rc
            return
        if args.host:
            rc = sess.tick_host_per_session(hfar, hnear, ms)[0] if args.classes > 1 else sess.tick_host(hfar, hnear, 40)[0]
            assert rc == 0, rc
            return
        off = (i % 8) * n * 2
        if args.asynchronous:
            rc = sess.tick_async(far.data_ptr() + off, near.data_ptr() + off, out.data_ptr(), far.shape[1], n, 40,
                                 ms_per_session=ms if args.classes > 1 else None)
        elif args.classes > 1:
            rc = sess.tick_device_per_session(far.data_ptr() + off, near.data_ptr() + off, out.data_ptr(), far.shape[1], n, ms)
        else:
            rc = sess.tick_device(far.data_ptr() + off, near.data_ptr() + off, out.data_ptr(), far.shape[1], n, 40)
        assert rc == 0, rc
    for i in range(40):                               # through the start-up phase
        tick(i)
    assert sess.synchronize() == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.ticks):
        tick(40 + i)
    assert sess.synchronize() == 0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.ticks
    blocks_per_tick = n / 64.0
    print(json.dumps({"streams": S, "fs": fs, "audio": "pinned host (zero copy)" if args.pinned else "host" if args.host else "device", "async": bool(args.asynchronous), "ms_per_tick": dt * 1e3, "frames_per_s": S * blocks_per_tick / dt,
                      "realtime_streams_per_gpu": int(S * 0.010 / dt)}))


if __name__ == "__main__":
    main()
