#!/usr/bin/env python3
"""Serving-shaped measurement (run ON THE GPU BOX): S concurrent WebRtcAecm_* sessions on a 10 ms clock
(WebRtcAecmSessions_Tick*).  Reports the time per tick, the bytes that cross the host boundary per second and how many
real-time streams one GPU sustains.  Not the headline metric (bench.py is); rows for BASELINE.md / INTEGRATION.md.

Where the audio lives (--audio):
  device           far / near / out rows in HBM (a media pipeline whose decoder and encoder run on the GPU too)
  host             pageable host memory through WebRtcAecmSessions_TickHost (staged by the library, synchronous)
  host-registered  caller-owned host buffers registered once (WebRtcAecmBatch_RegisterHostBuffer): the tick kernel reads
                   and writes them in place over the link -- no copies at all
  host-staged      pinned host buffers, three slots in flight: upload of tick t + 1, tick t and download of tick t - 1
                   overlap on three HIP streams, ordered by events only (WebRtcAecmSessions_TickAsync's wait / done hooks;
                   reference call shape: main.cc:112-145 -- far and near frames in, one output frame out per 10 ms)
A tick moves 2 x n x 2 bytes in and n x 2 bytes out per session (n = 160 at 16 kHz): 63 MB per tick of 65 536 sessions.
"""
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
    ap.add_argument("--audio", choices=["device", "host", "host-registered", "host-staged"], default="device")
    ap.add_argument("--host", action="store_true", help="= --audio host")
    ap.add_argument("--pinned", action="store_true", help="= --audio host-registered")
    ap.add_argument("--async", dest="asynchronous", action="store_true",
                    help="WebRtcAecmSessions_TickAsync: ticks are enqueued back to back, one synchronisation at the end (host-staged always is)")
    args = ap.parse_args()
    if args.host:
        args.audio = "host"
    if args.pinned:
        args.audio = "host-registered"
    import numpy as np
    import torch

    import webrtc_aecm_amd as aecm
    S, fs = args.streams, args.fs
    n = fs // 100                                     # one 10 ms tick
    g = torch.Generator(device="cuda").manual_seed(1)
    far = (torch.randn((S, n * 8), generator=g, device="cuda") * 3000).clamp_(-32768, 32767).to(torch.int16)
    near = (far.roll(37, dims=1) // 3 + (torch.randn((S, n * 8), generator=g, device="cuda") * 200).to(torch.int16))
    out = torch.empty_like(far)                       # Tick() uses ONE row stride for far, near and out
    sess = aecm.AecmSessions(S, fs, 1, 1)
    lib = aecm.load()
    torch.cuda.synchronize()
    ms = (40 + (np.arange(S) % args.classes)).astype(np.int16)          # distinct values in [40, 40 + classes)

    if args.audio in ("host", "host-registered"):
        hfar = np.ascontiguousarray(far.cpu().numpy()[:, :n])
        hnear = np.ascontiguousarray(near.cpu().numpy()[:, :n])
    if args.audio == "host-registered":
        hout = np.zeros_like(hnear)
        pf, pn, po = (aecm.register_host_buffer(a) for a in (hfar, hnear, hout))
    if args.audio == "host-staged":
        K = 3
        hfar_t, hnear_t = far[:, :n].contiguous().cpu().pin_memory(), near[:, :n].contiguous().cpu().pin_memory()
        hout_t = [torch.zeros((S, n), dtype=torch.int16).pin_memory() for _ in range(K)]
        dfar = [torch.zeros((S, n), dtype=torch.int16, device="cuda") for _ in range(K)]
        dnear = [torch.zeros_like(dfar[0]) for _ in range(K)]
        dout = [torch.zeros_like(dfar[0]) for _ in range(K)]
        upload, download = torch.cuda.Stream(), torch.cuda.Stream()
        ready = [torch.cuda.Event() for _ in range(K)]
        done = [torch.cuda.Event() for _ in range(K)]
        consumed = [None] * K
        for e in done:                                # torch creates the hipEvent_t at the first record; the tick re-records it
            e.record(download)
        torch.cuda.synchronize()

    def tick(i):
        if args.audio == "host-staged":
            k = i % K
            with torch.cuda.stream(upload):           # tick i's rows: host -> slot k, once the download of the tick that last used the slot is through
                if consumed[k] is not None:
                    upload.wait_event(consumed[k])
                dfar[k].copy_(hfar_t, non_blocking=True)
                dnear[k].copy_(hnear_t, non_blocking=True)
                ready[k].record(upload)
            rc = lib.WebRtcAecmSessions_TickAsync(sess.h, dfar[k].data_ptr(), dnear[k].data_ptr(), None, dout[k].data_ptr(), n, n, 40, None, None, None,
                                                  ready[k].cuda_event, done[k].cuda_event)
            assert rc == 0, rc
            with torch.cuda.stream(download):
                download.wait_event(done[k])
                hout_t[k].copy_(dout[k], non_blocking=True)
                consumed[k] = torch.cuda.Event()
                consumed[k].record(download)
            return
        if args.audio == "host-registered":
            rc = sess.tick_async(pf, pn, po, n, n, 40) if args.asynchronous else sess.tick_device(pf, pn, po, n, n, 40)
            assert rc == 0, rc
            return
        if args.audio == "host":
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
    boundary_bytes = 0 if args.audio == "device" else 3 * S * n * 2
    desc = aecm.describe_tick(S, aecm.device_info(0)[1])
    print(json.dumps({"streams": S, "fs": fs, "audio": args.audio, "async": bool(args.asynchronous or args.audio == "host-staged"),
                      "ms_per_tick": dt * 1e3, "frames_per_s": S * blocks_per_tick / dt, "realtime_streams_per_gpu": int(S * 0.010 / dt),
                      "MB_over_the_boundary_per_tick": boundary_bytes / 1e6, "GBps_over_the_boundary": boundary_bytes / dt / 1e9,
                      "tick_workgroup_rounds": desc["rounds_x1000"] / 1000.0}))


if __name__ == "__main__":
    main()
