#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED reference (oracle/_ref/libaecm_ref.so).

Run in the build container (needs /root/reference to build the reference .so).  The fixtures hold
only data: generator recipe ids (seed, n_blocks, fs, config) + expected outputs + state digests.
Inputs are regenerated bit-identically by webrtc_aecm_amd.synth on any machine.
"""
import hashlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import pyoracle  # noqa: E402
from webrtc_aecm_amd.synth import synth_clean, synth_pair  # noqa: E402

GOLD = ROOT / "tests" / "golden"

# (seed, n_blocks, fs, cng, echo_mode, profile)
BLOCK_CASES = [
    (0, 1200, 16000, 1, 3, None),
    (1, 1200, 16000, 1, 1, None),
    (2, 1200, 8000, 1, 4, None),
    (3, 1200, 8000, 0, 0, None),
    (100, 4200, 16000, 1, 3, "silent"),
]
# (seed, seconds, fs, frame, cng, echo_mode, ms, with nearendClean = synth_clean(near))
SESSION_CASES = [
    (7, 10, 16000, 160, 1, 1, 40, 0),     # the reference CLI's parameters (main.cc:102-105,163-164)
    (8, 6, 8000, 80, 1, 3, 40, 0),
    (9, 4, 16000, 160, 0, 2, 700, 0),     # ms out of range -> warning path, clamped
    (10, 6, 16000, 160, 1, 3, 40, 1),     # noise suppressor upstream: WebRtcAecm_Process(noisy, clean, ...)
    (11, 4, 8000, 160, 1, 1, 40, 1),
]
# Jittering msInSndCardBuf + far-end underruns (tests/helpers.py: call_pattern): (seed, seconds, fs, frame, cng, echo_mode)
SESSJIT_CASES = [
    (12, 6, 16000, 160, 1, 3),
    (13, 6, 8000, 80, 1, 1),
    (14, 5, 16000, 80, 0, 4),
]
# Far-end bursts (k = 0..3 and 30 WebRtcAecm_BufferFarend calls per WebRtcAecm_Process) + mid-session set_config / InitEchoPath /
# re-Init at the other rate (tests/helpers.py: call_pattern(bursts=True), reconfiguration_events): (seed, seconds, fs, frame)
SESSBURST_CASES = [
    (15, 3, 16000, 160),
    (16, 3, 8000, 80),
]


def main():
    pyoracle.build()
    assert pyoracle.have_reference(), "reference .so missing"
    GOLD.mkdir(parents=True, exist_ok=True)
    only = sys.argv[1] if len(sys.argv) > 1 else ""      # regenerate only the fixtures whose name contains this
    for seed, nb, fs, cng, em, prof in BLOCK_CASES:
        if only and only not in f"block_s{seed}_":
            continue
        far, near = synth_pair(seed, nb, fs, prof)
        r = pyoracle.RefCoreStream(fs, cng, em)
        digs = []
        outs = []
        for c in range(0, nb, 300):
            outs.append(r.process(far[c * 64:(c + 300) * 64], near[c * 64:(c + 300) * 64]))
            digs.append(r.digest())
        out = np.concatenate(outs)
        keep = out if prof is None else out[-600 * 64:]      # long cases: keep the tail + digests only
        np.savez_compressed(GOLD / f"block_s{seed}_fs{fs}_c{cng}_e{em}.npz", seed=seed, n_blocks=nb, fs=fs, cng=cng,
                            echo_mode=em, profile=prof or "", out=keep, digests=np.stack(digs),
                            sha256=hashlib.sha256(out.tobytes()).hexdigest())
        print("block", seed, fs, cng, em, prof, hashlib.sha256(out.tobytes()).hexdigest()[:16])
    for seed, secs, fs, frame, cng, em, ms, with_clean in SESSION_CASES:
        name = f"session_s{seed}_fs{fs}_f{frame}_c{cng}_e{em}_ms{ms}" + ("_clean" if with_clean else "")
        if only and only not in name:
            continue
        nb = secs * fs // 64
        far, near = synth_pair(seed, nb, fs, "mixed")
        clean = synth_clean(near) if with_clean else None
        n = (far.size // frame) * frame
        s = pyoracle.RefSession(fs, cng, em)
        lib = s.lib
        out = near.copy()
        codes = set()
        buf = np.empty(frame, dtype=np.int16)
        for i in range(n // frame):
            f = far[i * frame:(i + 1) * frame]
            d = out[i * frame:(i + 1) * frame]
            c = clean[i * frame:(i + 1) * frame].ctypes.data if with_clean else None
            assert lib.WebRtcAecm_BufferFarend(s.h, f.ctypes.data, frame) == 0
            codes.add(int(lib.WebRtcAecm_Process(s.h, d.ctypes.data, c, buf.ctypes.data, frame, ms)))
            d[:] = buf
        np.savez_compressed(GOLD / f"{name}.npz", seed=seed, n_blocks=nb,
                            fs=fs, frame=frame, cng=cng, echo_mode=em, ms=ms, clean=with_clean, out=out[:n],
                            codes=np.array(sorted(codes)), sha256=hashlib.sha256(out[:n].tobytes()).hexdigest())
        print("session", seed, fs, frame, cng, em, ms, with_clean, sorted(codes))
    sys.path.insert(0, str(ROOT / "tests"))
    from helpers import call_pattern, drive_session
    for seed, secs, fs, frame, cng, em in SESSJIT_CASES:
        name = f"sessjit_s{seed}_fs{fs}_f{frame}_c{cng}_e{em}"
        if only and only not in name:
            continue
        nb = secs * fs // 64
        far, near = synth_pair(seed, nb, fs, "mixed")
        n_calls = far.size // frame
        ms_seq, far_present = call_pattern(seed, n_calls)
        out, codes = drive_session(pyoracle.RefSession(fs, cng, em), far, near, frame, ms_seq, far_present)
        np.savez_compressed(GOLD / f"{name}.npz", seed=seed, n_blocks=nb, fs=fs, frame=frame, cng=cng, echo_mode=em,
                            ms_seq=ms_seq, far_present=far_present, out=out, codes=codes,
                            sha256=hashlib.sha256(out.tobytes()).hexdigest())
        print("sessjit", seed, fs, frame, cng, em, sorted(set(codes.tolist())), int((far_present == 0).sum()), "underruns")
    from helpers import far_frames_needed, reconfiguration_events
    for seed, secs, fs, frame in SESSBURST_CASES:
        name = f"sessburst_s{seed}_fs{fs}_f{frame}"
        if only and only not in name:
            continue
        n_calls = secs * fs // frame
        ms_seq, far_calls = call_pattern(seed, n_calls, bursts=True)
        far, _ = synth_pair(seed, far_frames_needed(far_calls) * frame // 64 + 1, fs, "mixed")
        _, near = synth_pair(seed, n_calls * frame // 64 + 1, fs, "mixed")
        r = pyoracle.RefSession(fs, 1, 3)
        out, codes = drive_session(r, far, near, frame, ms_seq, far_calls, events=reconfiguration_events(fs, n_calls))
        np.savez_compressed(GOLD / f"{name}.npz", seed=seed, n_calls=n_calls, fs=fs, frame=frame, ms_seq=ms_seq, far_calls=far_calls, out=out,
                            codes=codes, paths=np.stack(r.event_log), sha256=hashlib.sha256(out.tobytes()).hexdigest())
        print("sessburst", seed, fs, frame, sorted(set(codes.tolist())), int(far_calls.sum()), "far calls for", n_calls, "near calls")
    if only:
        return
    # 60 s reference-CLI-shaped run: hash only
    far, near = synth_pair(60, 15000, 16000, "mixed")
    s = pyoracle.RefSession(16000, 1, 1)
    out = s.run(far, near, 160, 40)
    (GOLD / "session_60s_16k.sha256").write_text(hashlib.sha256(out.tobytes()).hexdigest() + "\n")
    print("60 s hash", hashlib.sha256(out.tobytes()).hexdigest()[:16])


if __name__ == "__main__":
    main()
