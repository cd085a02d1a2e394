#!/usr/bin/env python3
"""PCIe-inclusive rate of the block path with the audio in HOST memory: WebRtcAecmBatch_ProcessBlocksHost (far / near in
pageable host memory in, out in pageable host memory out; staged through device buffers), and with --registered the
zero-copy form (caller-owned buffers registered once with WebRtcAecmBatch_RegisterHostBuffer, the kernel reads and writes
them in place over the link).  Not the headline metric (bench.py keeps the audio resident in HBM); rows for DESIGN.md."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=65536)
    ap.add_argument("--blocks", type=int, default=128)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--registered", action="store_true", help="zero-copy: registered (pinned + mapped) host buffers, ProcessBlocks on their device aliases")
    a = ap.parse_args()
    import webrtc_aecm_amd as aecm
    S, T = a.streams, a.blocks
    rs = np.random.RandomState(1)
    far = (rs.standard_normal((S, T * 64)).astype(np.float32) * 3000).clip(-32768, 32767).astype(np.int16)
    near = (np.roll(far, 37, axis=1) // 3).astype(np.int16)
    b = aecm.AecmBatch(S, 16000, 1, 1)
    out = np.zeros_like(near)                                   # touched once: no page faults inside the timed calls
    if a.registered:
        ptrs = [aecm.register_host_buffer(x) for x in (far, near, out)]

        def call():
            b.process_device(ptrs[0], ptrs[1], ptrs[2], T * 64, 64, T)
            b.synchronize()
    else:
        def call():
            b.process_host(far, near, out=out)
    call()                                                      # warm-up (device allocations)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        call()
    dt = (time.perf_counter() - t0) / a.reps
    frames = S * T
    print(json.dumps({"streams": S, "blocks": T, "host_memory": "registered (zero copy)" if a.registered else "pageable (staged)",
                      "s_per_call": dt, "frames_per_s": frames / dt, "GBps_over_the_boundary": frames * 384 / dt / 1e9}))
    if a.registered:
        for x in (far, near, out):
            aecm.unregister_host_buffer(x)


if __name__ == "__main__":
    main()
