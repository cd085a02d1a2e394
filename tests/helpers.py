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


def checker_name(prefer_reference=True):
    """Which CPU checker the full-size tests use: the unmodified reference when its prebuilt library travelled with the tree
    (oracle/_ref), else the restatement (itself pinned to the reference by tests/test_oracle.py where the reference exists)."""
    return "reference" if prefer_reference and pyoracle.have_reference() else "port"


def oracle_batch(seeds, n_blocks, fs, configs, workers=None, pairs=None, prefer_reference=False):
    """CPU checker over many streams in parallel threads (ctypes releases the GIL).  pairs = (far, near) [len(seeds), L]
    arrays already synthesised from these seeds (saves synthesising them a second time).  prefer_reference: run the
    UNMODIFIED REFERENCE (pyoracle.RefCoreStream: WebRtcAecm_ProcessBlock itself) instead of our restatement of it when
    oracle/_ref is there -- the HIP path is then compared with the reference directly, not through the restatement."""
    use_ref = checker_name(prefer_reference) == "reference"

    def one(i):
        cng, em = configs[i]
        if pairs is not None or use_ref:
            far, near = (pairs[0][i], pairs[1][i]) if pairs is not None else synth_pair(seeds[i], n_blocks, fs)
            o = pyoracle.RefCoreStream(fs, cng, em) if use_ref else pyoracle.OracleStream(fs, cng, em)
            return o.process(far, near), o.digest()
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


def adversarial_cases(n_cases=24, n_blocks=1000, seed=7):
    """Deterministic hostile inputs for the block path: full-scale noise, +-full-scale square waves,
    constants (incl. -32768), sparse spikes, LSB dither, full-scale tones, and echo-like mixtures of
    them; with a random configuration and (every other case) a random full-range echo path.
    Yields dicts: far, near, fs, cng, echo_mode, path (int16[65] or None)."""
    rs = np.random.RandomState(seed)

    def nasty(n, kind):
        if kind == 0:
            return rs.randint(-32768, 32768, size=n).astype(np.int16)
        if kind == 1:
            return np.where(rs.randint(0, 2, size=n) == 1, 32767, -32768).astype(np.int16)
        if kind == 2:
            return np.full(n, rs.choice([-32768, 32767, 1, -1, 0, 16384]), dtype=np.int16)
        if kind == 3:
            x = np.zeros(n, dtype=np.int16)
            idx = rs.randint(0, n, size=n // 50)
            x[idx] = rs.randint(-32768, 32768, size=idx.size)
            return x
        if kind == 4:
            return rs.randint(-3, 4, size=n).astype(np.int16)
        t = np.arange(n)
        return (32767 * np.sin(2 * np.pi * t * rs.randint(1, 60) / 128.0)).astype(np.int16)

    for it in range(n_cases):
        n = n_blocks * 64
        far, near = nasty(n, rs.randint(0, 6)), nasty(n, rs.randint(0, 6))
        if it % 3 == 0:
            near = np.clip(np.roll(far.astype(np.int32), rs.randint(0, 2000)) // rs.choice([1, 2, 8, 64]) +
                           near // rs.choice([1, 4, 64, 1024]), -32768, 32767).astype(np.int16)
        path = None
        if it % 2 == 1:
            path = rs.randint(-32768, 32768, size=65).astype(np.int16) if it % 4 == 1 else \
                rs.choice([0, 1, -1, 32767, -32768, 12000], size=65).astype(np.int16)
        yield dict(far=far, near=near, fs=int(rs.choice([8000, 16000])), cng=int(rs.randint(0, 2)),
                   echo_mode=int(rs.randint(0, 5)), path=path)


def call_pattern(seed, n_calls, bursts=False):
    """A hostile but deterministic call pattern for session tests: msInSndCardBuf jitters around 40 ms with
    occasional out-of-range / large excursions, and now and then a call comes without a WebRtcAecm_BufferFarend
    (far-end underrun).  Returns (ms[int16 n_calls], far_calls[uint8 n_calls]): far_calls[i] = the number of
    WebRtcAecm_BufferFarend calls before the i-th WebRtcAecm_Process -- 0 or 1 by default; with bursts=True what a
    jittery network delivers: k = 0, 1, 1, 1, 2, 3 in turn (rotated by the seed), a 30-frame burst every 50 calls (the
    jitter buffer overflows: reference ring_buffer.c:142-170) and, once, 12 calls without any far frame."""
    rs = np.random.RandomState(1000 + seed)
    ms = (40 + rs.randint(-12, 13, size=n_calls)).astype(np.int16)
    idx = np.arange(n_calls)
    special = np.nonzero(idx % 40 == 39)[0]
    ms[special] = rs.choice([-3, 0, 600, 90, 250], size=special.size)
    far_calls = np.ones(n_calls, dtype=np.uint8)
    far_calls[idx % 97 == 96] = 0
    far_calls[(idx % 211 >= 205) & (idx > 100)] = 0                                        # a run of underruns
    if bursts:
        far_calls = np.array([0, 1, 1, 1, 2, 3], dtype=np.uint8)[(idx + seed) % 6]
        far_calls[idx % 50 == (49 - seed % 7)] = 30
        far_calls[(idx >= 120) & (idx < 132)] = 0
    return ms, far_calls


def reconfiguration_events(fs, n_calls):
    """Mid-session control calls for drive_session(events=...), the same for every ABI-shaped session object:
    WebRtcAecm_set_config (valid and refused), InitEchoPath, GetEchoPath, and WebRtcAecm_Init at the OTHER rate and back
    (reference echo_control_mobile.cc:142-191, 410-532).  Every return code is checked against the reference's; the echo
    paths read on the way are appended to sess.event_log for the caller to compare."""
    other = 8000 if fs == 16000 else 16000
    path = (np.arange(65) * 97 % 5000 + 50).astype(np.int16)

    def expect(rc, want):
        assert rc == want, (rc, want)

    def read_path(sess):                       # what the session has made of the path by now: compared between the sessions by the caller
        rc, p = sess.get_echo_path()
        assert rc == 0
        sess.__dict__.setdefault("event_log", []).append(p.copy())
    return {
        n_calls // 5: lambda sess: expect(sess.set_config(0, 4), 0),
        n_calls // 4: lambda sess: expect(sess.set_config(1, 7), 12004),                 # refused: echoMode out of range (cngMode is committed first)
        n_calls // 3: lambda sess: expect(sess.init_echo_path(path), 0),
        n_calls // 3 + 1: read_path,
        n_calls // 3 + 40: read_path,
        n_calls - 1: read_path,
        n_calls // 2: lambda sess: (expect(sess.init(other), 0), expect(sess.set_config(1, 2), 0)),
        (2 * n_calls) // 3: lambda sess: expect(sess.init(12345), 12004),                # refused: the session keeps running as it was
        (3 * n_calls) // 4: lambda sess: expect(sess.init(fs), 0),                       # back: default configuration again
    }


def far_frames_needed(far_calls):
    """How many far frames a session driven with this far_calls sequence consumes."""
    return int(np.asarray(far_calls, dtype=np.int64).sum())


def drive_session(sess, far, near, frame, ms_seq, far_calls=None, clean=None, events=None):
    """Drive an ABI-shaped session object (webrtc_aecm_amd.Aecm, pyoracle.RefSession, simlib.SimSession) call by
    call: before the i-th WebRtcAecm_Process, far_calls[i] WebRtcAecm_BufferFarend calls (one when far_calls is None),
    each taking the next `frame` samples of far.  events: {call index: function(sess)} run before that call's far
    frames (mid-session WebRtcAecm_set_config / InitEchoPath / Init ...).  Returns (out, codes[n_calls])."""
    n_calls = near.size // frame
    out = np.empty(n_calls * frame, dtype=np.int16)
    codes = np.zeros(n_calls, dtype=np.int32)
    cursor = 0
    # 0 / 1 patterns keep far and near aligned: the far frame of an underrun is lost, not delayed
    aligned = far_calls is not None and int(np.asarray(far_calls).max(initial=0)) <= 1
    for i in range(n_calls):
        sl = slice(i * frame, (i + 1) * frame)
        if events and i in events:
            events[i](sess)
        for _ in range(1 if far_calls is None else int(far_calls[i])):
            rc = sess.buffer_farend(far[cursor:cursor + frame])
            assert rc == 0, (i, rc)
            cursor += frame
        if aligned and not far_calls[i]:
            cursor += frame
        codes[i], out[sl] = sess.process(near[sl], None if clean is None else clean[sl], int(ms_seq[i]))
    return out, codes


def run_burst_fixture(sess, g):
    """Drive an ABI-shaped session through a sessburst_* fixture (tools/gen_golden.py); returns (out, codes, echo paths read)."""
    fs, frame, n_calls, seed = int(g["fs"]), int(g["frame"]), int(g["n_calls"]), int(g["seed"])
    far, _ = synth_pair(seed, far_frames_needed(g["far_calls"]) * frame // 64 + 1, fs, "mixed")
    _, near = synth_pair(seed, n_calls * frame // 64 + 1, fs, "mixed")
    assert sess.init(fs) == 0 and sess.set_config(1, 3) == 0
    out, codes = drive_session(sess, far, near, frame, g["ms_seq"], g["far_calls"], events=reconfiguration_events(fs, n_calls))
    return out, codes, np.stack(sess.event_log)


# ---- WAV files in the sample formats the reference CLI reads (dr_wav) ----------------------------------------------------
# "s16_align4": a 16-bit file whose header claims a block alignment of 4 -- dr_wav goes by the bit depth (dr_wav.h:1815-1827)
WAV_FORMATS = ("u8", "s16", "s24", "s32", "s16_ext", "s24_ext", "f32", "f64", "f32_ext", "alaw", "mulaw", "s16_align4")


def write_wav_format(path, rate, x, fmt):
    """x: int16 samples.  Writes them in `fmt` (from WAV_FORMATS): values are spread over each format's range in a way that
    exercises its rounding (float formats get non-representable fractions, 24/32-bit formats non-zero low bytes,
    the companded formats every code)."""
    import struct
    x = np.asarray(x, dtype=np.int64)
    rs = np.random.RandomState(len(x) * 31 + len(fmt))
    ext = fmt.endswith("_ext")
    base = fmt[:-4] if ext else fmt
    bad_align = fmt == "s16_align4"
    if bad_align:
        base = "s16"
    if base == "u8":
        tag, bits, data = 1, 8, (((x >> 8) + 128) & 0xff).astype(np.uint8).tobytes()
    elif base == "s16":
        tag, bits, data = 1, 16, x.astype("<i2").tobytes()
    elif base == "s24":
        v = (x << 8) + rs.randint(0, 256, size=x.size)
        b = np.empty((x.size, 3), np.uint8)
        b[:, 0], b[:, 1], b[:, 2] = v & 0xff, (v >> 8) & 0xff, (v >> 16) & 0xff
        tag, bits, data = 1, 24, b.tobytes()
    elif base == "s32":
        tag, bits, data = 1, 32, ((x << 16) + rs.randint(0, 65536, size=x.size)).astype("<i4").tobytes()
    elif base in ("f32", "f64"):
        v = x / 32768.0 + rs.uniform(-0.4, 0.4, size=x.size) / 32768.0          # off the int16 grid: the rounding rule matters
        v[:8] = [-1.0, 1.0, -1.5, 1.5, 0.0, -0.0, 0.99999, -0.99999][: min(8, x.size)]
        tag, bits, data = 3, (32 if base == "f32" else 64), v.astype("<f4" if base == "f32" else "<f8").tobytes()
    elif base in ("alaw", "mulaw"):
        codes = ((x >> 8) & 0xff).astype(np.uint8)
        codes[:256] = np.arange(min(256, x.size), dtype=np.uint8)                 # every code once
        tag, bits, data = (6 if base == "alaw" else 7), 8, codes.tobytes()
    else:
        raise ValueError(fmt)
    align = 4 if bad_align else bits // 8
    if ext:
        guid_tail = bytes.fromhex("000000001000800000aa00389b71")
        fmt_chunk = struct.pack("<HHIIHHHHI", 0xFFFE, 1, rate, rate * align, align, bits, 22, bits, 4) + struct.pack("<H", tag) + guid_tail
    else:
        fmt_chunk = struct.pack("<HHIIHH", tag, 1, rate, rate * align, align, bits)
        if tag != 1:
            fmt_chunk += struct.pack("<H", 0)                                       # cbSize of the non-PCM header
    chunks = b"fmt " + struct.pack("<I", len(fmt_chunk)) + fmt_chunk
    if tag != 1 and not ext:
        chunks += b"fact" + struct.pack("<II", 4, len(data) // align)
    chunks += b"LIST" + struct.pack("<I", 5) + b"INFOx" + b"\0"                     # an odd-sized chunk to skip (padded)
    chunks += b"data" + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b"")
    Path(path).write_bytes(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)
