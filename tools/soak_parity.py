#!/usr/bin/env python3
"""Every frame of a full-size pass against the unmodified reference (run ON THE GPU BOX).

bench.py proves its timed workload on 18 streams; the parity tests run thousands of streams at test sizes.  This tool
closes the gap once per kernel change that cannot be reviewed line by line (a compiler flag, a new data-dependent path):
it builds bench.py's own batch (S streams x T blocks, same generator and seed), runs PASSES launches on the GPU, then
pushes EVERY stream through the CPU checker (oracle/_ref, the reference built from its own sources; our restatement if
that library did not travel) on all usable host cores and compares every output sample of the last pass and every
stream's final state digest.  ~50 s of CPU per pass of 65 536 x 1 280 frames on 16 cores.

    python tools/soak_parity.py [--streams 65536] [--blocks 1280] [--passes 2] [--fs 16000] [--clean] [--echo-mode 1] [--cng 1]
"""
import argparse, json, sys, time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=65536)
    ap.add_argument("--blocks", type=int, default=1280)
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--fs", type=int, default=16000)
    ap.add_argument("--cng", type=int, default=1)
    ap.add_argument("--echo-mode", type=int, default=1)
    ap.add_argument("--clean", action="store_true")
    ap.add_argument("--variant", choices=["fast", "safe"], default="fast")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--chunk", type=int, default=4096, help="streams checked per host round trip")
    a = ap.parse_args()

    import torch
    import bench
    import webrtc_aecm_amd as aecm
    from webrtc_aecm_amd import build as _build
    from oracle import pyoracle
    use_ref = pyoracle.have_reference()
    device = torch.device("cuda", 0)
    S, T = a.streams, a.blocks
    far, near = bench.synth_on_device(torch, S, T * 64, a.seed, device)
    clean = (near.to(torch.int32) * 3 // 4).to(torch.int16) if a.clean else None
    batch = aecm.AecmBatch(S, a.fs, cng_mode=a.cng, echo_mode=a.echo_mode, device=0,
                           variant=aecm.KERNEL_FAST if a.variant == "fast" else aecm.KERNEL_SAFE)
    out = torch.empty_like(near)
    for _ in range(a.passes):
        batch.process_device(far.data_ptr(), near.data_ptr(), out.data_ptr(), far.shape[1], 64, T,
                             clean.data_ptr() if clean is not None else None)
    batch.synchronize()
    torch.cuda.synchronize()

    cores = bench.usable_cores()
    bad_streams, bad_samples, bad_digests = [], 0, 0
    t0 = time.perf_counter()
    for c0 in range(0, S, a.chunk):
        c1 = min(S, c0 + a.chunk)
        f, d, got = far[c0:c1].cpu().numpy(), near[c0:c1].cpu().numpy(), out[c0:c1].cpu().numpy()
        cl = clean[c0:c1].cpu().numpy() if clean is not None else None
        digests = [batch.digest(i) for i in range(c0, c1)]

        def one(k):
            chk = pyoracle.RefCoreStream(a.fs, a.cng, a.echo_mode) if use_ref else pyoracle.OracleStream(a.fs, a.cng, a.echo_mode)
            exp = None
            for _ in range(a.passes):
                if cl is None:
                    exp = chk.process(f[k], d[k])
                else:
                    exp = np.concatenate([chk.process_block_clean(f[k][b * 64:(b + 1) * 64], d[k][b * 64:(b + 1) * 64],
                                                                  cl[k][b * 64:(b + 1) * 64]) for b in range(T)])
            return int(np.count_nonzero(exp != got[k])), bool(np.array_equal(chk.digest(), digests[k]))
        with ThreadPoolExecutor(max_workers=cores) as ex:
            for k, (nbad, dig_ok) in enumerate(ex.map(one, range(c1 - c0))):
                if nbad or not dig_ok:
                    bad_streams.append(c0 + k)
                    bad_samples += nbad
                    bad_digests += not dig_ok
    line = {"what": "every stream of a full-size batch (bench.py's generator) on the CPU checker: last pass's output samples and final state "
                    "digests compared bit for bit",
            "checker": "reference" if use_ref else "port", "streams": S, "blocks_per_pass": T, "passes": a.passes, "fs": a.fs,
            "cng": a.cng, "echo_mode": a.echo_mode, "clean_input": bool(a.clean), "variant": a.variant, "seed": a.seed,
            "frames_processed_per_side": S * T * a.passes, "samples_compared": S * T * 64, "digests_compared": S,
            "ok": not bad_streams, "mismatching_streams": bad_streams[:32], "mismatching_samples": bad_samples,
            "mismatching_digests": bad_digests, "cpu_cores": cores, "cpu_seconds_wall": round(time.perf_counter() - t0, 1),
            "launch_form": dict(zip(("form", "chunk_blocks"), batch.describe_launch(T, bool(a.clean)))),      # 2 = chunk queue (include/aecm_batch.h)
            "library": str(aecm.library_path()), "build": _build.build_info()}
    print(json.dumps(line))
    return 0 if not bad_streams else 1


if __name__ == "__main__":
    sys.exit(main())
