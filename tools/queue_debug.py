"""Run ON THE GPU BOX: the chunk-queue kernel on small batches, step by step, with wall times (debugging aid)."""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import webrtc_aecm_amd as aecm
from helpers import synth_streams
from oracle import pyoracle

def run(S, T, chunk, min_streams, fs=16000):
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
    t0 = time.time()
    try:
        out = b.process_host(far, near)
    except Exception as e:
        print(f"S={S} T={T} chunk={chunk}: FAILED after {time.time()-t0:.2f}s: {e}", flush=True)
        return
    dt = time.time() - t0
    bad = [s for s in range(S) if not np.array_equal(out[s], exp[s % len(seeds)][0])]
    badd = [s for s in range(S) if not np.array_equal(b.digest(s), exp[s % len(seeds)][1])][:5] if S <= 64 else []
    print(f"S={S} T={T} chunk={chunk}: {dt:.2f}s, output mismatches {len(bad)} {bad[:8]}, digest mismatches {badd}", flush=True)
    b.close()

run(24, 192, 0, 0)
run(4, 64, 16, 0)
run(24, 192, 64, 0)
run(24, 192, 4, 0)
run(2048, 192, 32, 0)
run(9001, 256, 128, -1)
run(9001, 256, 32, -1)
