"""Deterministic, integer-only synthetic far/near-end signal pairs for parity tests and bench.py.

The reference ships no audio and no tests, so inputs are *designed* to reach
the rare branches of the block path (recipe: SURVEY.md section 4):

  * far  = Gaussian-ish white noise x piecewise-constant envelope drawn per 0.4 s from
           {15, 60, 500, 3000, 9000, 20000} (far-energy dynamics > FAR_ENERGY_DIFF,
           reference aecm/aecm_defines.h:36), 3-tap smoothed;
  * near = far * sparse echo path, with (a) a very weak echo at the beginning (triggers the
           firstVAD channel>>3 rescue, reference aecm/aecm_core.cc:741-753), (b) an echo-path switch
           half-way (ResetAdaptiveChannel, aecm_core.cc:959), (c) near-end talk bursts (double
           talk), (d) a stretch of exact digital silence on near while far is active (low-level
           comfort-noise branches, aecm_core_c.cc:90-98,117-125), (e) a stretch of +/- full-scale
           near (noise clamp aecm_core_c.cc:131-134, output saturation :225), (f) stretches of exact
           zeros on far (far popcount 0 => delay-estimator freeze, delay_estimator.cc:558,628).

Everything is int64 numpy arithmetic on `RandomState.randint` draws (MT19937, frozen legacy
generator), so the same (seed, n_blocks, fs, profile) gives bit-identical int16 arrays on every
machine -- the GPU box regenerates the inputs instead of shipping them.
"""
from __future__ import annotations

import numpy as np

BLOCK = 64
_LEVELS = np.array([15, 60, 500, 3000, 9000, 20000], dtype=np.int64)
_PATH_A = ((100, 16384), (180, -9830), (333, 6553), (600, 3276))     # (delay samples, Q15 gain)
_PATH_B = ((60, 13107), (140, 11468), (410, -8192), (520, 4915))
PROFILES = ("mixed", "steady", "loud", "sparse")
EXTRA_PROFILES = ("silent",)   # near is exact digital silence for the first 85 % (noise-floor floor branches)


def _noise(rs: np.random.RandomState, n: int) -> np.ndarray:
    """Sum of four uniform ints: zero-mean, std ~ 4730, |x| < 16384 (int64)."""
    return rs.randint(-4096, 4096, size=(4, n)).astype(np.int64).sum(axis=0)


def _fir(x: np.ndarray, taps) -> np.ndarray:
    y = np.zeros_like(x)
    for d, g in taps:
        y[d:] += g * x[: len(x) - d]
    return y >> 15


def synth_pair(seed: int, n_blocks: int, fs: int = 16000, profile: str | None = None):
    """Return (far, near) int16 arrays of n_blocks*64 samples."""
    if profile is None:
        profile = PROFILES[seed % len(PROFILES)]
    if profile not in PROFILES + EXTRA_PROFILES:
        raise ValueError(profile)
    rs = np.random.RandomState((seed * 2654435761 + 12345) % (2 ** 32))
    n = n_blocks * BLOCK
    seg = int(0.4 * fs)
    if n < 40 * seg:                      # short runs: compress the schedule to ~40 segments
        seg = max(BLOCK, (n // 40) // BLOCK * BLOCK)
    nseg = n // seg + 2
    t = np.arange(n, dtype=np.int64)

    # ---- far end ----
    if profile == "steady":
        env = np.repeat(_LEVELS[rs.randint(2, 5, size=nseg)], seg)[:n]
    elif profile == "loud":
        env = np.repeat(_LEVELS[rs.randint(3, 6, size=nseg)] * 4, seg)[:n]
    else:
        env = np.repeat(_LEVELS[rs.randint(0, 6, size=nseg)], seg)[:n]
    w = _noise(rs, n)
    raw = (w * env) >> 13
    far = raw.copy()
    far[1:-1] = (raw[:-2] + 2 * raw[1:-1] + raw[2:]) >> 2
    if profile in ("mixed", "sparse"):
        # exact digital zeros on far for a few segments
        zero_seg = rs.randint(0, nseg, size=max(1, nseg // (6 if profile == "mixed" else 3)))
        mask = np.zeros(nseg * seg, dtype=bool)
        for z in zero_seg:
            mask[z * seg:(z + 1) * seg] = True
        far[mask[:n]] = 0
    far = np.clip(far, -32768, 32767)

    # ---- echo ----
    switch = n // 2 + int(rs.randint(-seg, seg))
    echo = np.where(t < switch, _fir(far, _PATH_A), _fir(far, _PATH_B))
    weak_until = int(n * 0.13)
    if profile in ("mixed", "sparse"):
        echo = np.where(t < weak_until, (echo * 328) >> 15, echo)     # gain 0.01
    if profile == "loud":
        echo = echo * 3

    # ---- near-end talk (double talk) ----
    lv = np.array([0, 0, 0, 2000, 8000], dtype=np.int64)
    talk_env = np.repeat(lv[rs.randint(0, 5, size=nseg)], seg)[:n]
    if profile == "steady":
        talk_env = np.repeat(np.array([0, 40], dtype=np.int64)[rs.randint(0, 2, size=nseg)], seg)[:n]
    talk = (_noise(rs, n) * talk_env) >> 12
    near = echo + talk

    if profile in ("mixed", "sparse"):
        s0, s1 = int(n * 0.62), int(n * 0.75)
        near[s0:s1] = 0                                               # exact silence on near
    if profile in ("mixed", "loud"):
        f0, f1 = int(n * 0.90), int(n * 0.93)
        sign = rs.randint(0, 2, size=f1 - f0).astype(np.int64)
        near[f0:f1] = np.where(sign == 1, 32767, -32768)
        r0 = f0 + (f1 - f0) // 2
        near[r0:r0 + 200] = -32768                                    # a run of -32768
    if profile == "silent":
        near[: int(n * 0.85)] = 0
    near = np.clip(near, -32768, 32767)
    return far.astype(np.int16), near.astype(np.int16)


def synth_batch(base_seed: int, n_streams: int, n_blocks: int, fs: int = 16000):
    """Stream-major [S, n_blocks*64] int16 arrays; stream s uses seed base_seed + s."""
    far = np.empty((n_streams, n_blocks * BLOCK), dtype=np.int16)
    near = np.empty_like(far)
    for s in range(n_streams):
        far[s], near[s] = synth_pair(base_seed + s, n_blocks, fs)
    return far, near


def synth_clean(near):
    """A deterministic "noise-suppressed" companion of a near-end signal (WebRtcAecm_Process's nearendClean)
    for tests: 3/4 of the amplitude, floor division, integer only."""
    return (np.asarray(near).astype(np.int32) * 3 // 4).astype(np.int16)
