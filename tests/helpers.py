"""Shared helpers for the parity tests."""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

from oracle import pyoracle
from webrtc_aecm_amd.synth import synth_pair

GOLDEN = Path(__file__).resolve().parent / "golden"


def stream_config(s: int):
    """Per-stream (cng_mode, echo_mode) used by the multi-stream parity tests."""
    return (0 if s % 7 == 3 else 1), s % 5


def oracle_run(seed, n_blocks, fs, cng, echo_mode, profile=None, chunks=None):
    """Run the CPU oracle on a synthetic stream; returns (out, digest)."""
    far, near = synth_pair(seed, n_blocks, fs, profile)
    o = pyoracle.OracleStream(fs, cng, echo_mode)
    out = o.process(far, near)
    return out, o.digest()


def oracle_batch(seeds, n_blocks, fs, configs, workers=None):
    """Oracle over many streams in parallel threads (ctypes releases the GIL)."""
    def one(i):
        cng, em = configs[i]
        return oracle_run(seeds[i], n_blocks, fs, cng, em)
    with ThreadPoolExecutor(max_workers=workers) as ex:
        res = list(ex.map(one, range(len(seeds))))
    return np.stack([r[0] for r in res]), np.stack([r[1] for r in res])


def synth_streams(seeds, n_blocks, fs):
    far = np.empty((len(seeds), n_blocks * 64), dtype=np.int16)
    near = np.empty_like(far)
    for i, s in enumerate(seeds):
        far[i], near[i] = synth_pair(s, n_blocks, fs)
    return far, near


def describe_digest_diff(a, b):
    return [pyoracle.DIGEST_NAMES[i] for i in np.nonzero(np.asarray(a) != np.asarray(b))[0]]


def golden_files(prefix):
    return sorted(GOLDEN.glob(prefix + "*.npz"))
