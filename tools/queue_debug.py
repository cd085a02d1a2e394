"""Run ON THE GPU BOX: the launch forms of the block kernels on small batches, step by step, with wall times (debugging aid).
    python tools/queue_debug.py queue|pipe"""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import webrtc_aecm_amd as aecm
from helpers import synth_streams
from oracle import pyoracle

def run(S, T, chunk=0, min_streams=0, pipe=0, fs=16000):
    seeds = list(range(7100, 7100 + min(S, 32)))
    far, near = synth_streams(seeds, T, fs)
    reps = (S + len(seeds) - 1) // len(seeds)
    far = np.tile(far, (reps, 1))[:S].copy(); near = np.tile(near, (reps, 1))[:S].copy()
    exp = []
    for k in range(len(seeds)):
        o = pyoracle.OracleStream(fs, 1, 3)
        exp.append((o.process(far[k], near[k]), o.digest()))
    b = aecm.AecmBatch(S, fs)
    b.set_launch_chunking(chunk, min_streams)
    b.set_launch_pipelining(pipe)
    form = b.describe_launch(T)
    t0 = time.time()
    try:
        half = (T // 2) * 64
        out = np.concatenate([b.process_host(far[:, :half], near[:, :half]), b.process_host(far[:, half:], near[:, half:])], axis=1)
    except Exception as e:
        print(f"S={S} T={T} form={form}: FAILED after {time.time()-t0:.2f}s: {e}", flush=True)
        return
    dt = time.time() - t0
    bad = [s for s in range(S) if not np.array_equal(out[s], exp[s % len(seeds)][0])]
    badd = [s for s in range(S) if not np.array_equal(b.digest(s), exp[s % len(seeds)][1])][:5] if S <= 512 else []
    print(f"S={S} T={T} form={form}: {dt:.2f}s, output mismatches {len(bad)} {bad[:8]}, digest mismatches {badd}", flush=True)
    b.close()

what = sys.argv[1] if len(sys.argv) > 1 else "queue"
if what == "queue":
    run(24, 192)
    run(4, 64, 16)
    run(24, 192, 64)
    run(24, 192, 4)
    run(2048, 192, 32)
    run(9001, 256, 128, -1)
else:
    for S in (1, 2, 3, 4, 5, 7, 8, 64, 333, 4096):
        run(S, 192, pipe=1)
    run(8, 192, pipe=1, fs=8000)
    run(4, 1, pipe=1)
    run(4, 3, pipe=1)
