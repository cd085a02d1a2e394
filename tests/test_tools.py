"""tools/first_divergence.py, the parity triage tool (the reference's AEC_DEBUG dump analogue), on the CPU lane simulator:
it must report nothing on an intact build and the exact block when one side's input is perturbed."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
TOOL = ROOT / "tools" / "first_divergence.py"


def _run(*args):
    return subprocess.run([sys.executable, str(TOOL), *args], capture_output=True, text=True, timeout=900)


def test_first_divergence_reports_nothing_then_the_perturbed_block():
    r = _run("--engine", "sim", "--blocks", "260", "--seed", "11", "--fs", "8000", "--echo-mode", "1")
    assert r.returncode == 0 and "no divergence in 260 blocks" in r.stdout, r.stdout + r.stderr
    r = _run("--engine", "sim", "--blocks", "260", "--seed", "11", "--perturb-block", "97")
    assert r.returncode == 1 and "FIRST DIVERGENCE at block 97 " in r.stdout, r.stdout + r.stderr
    assert "H(xBuf,dBuf,outBuf,far_history)" in r.stdout          # the perturbed near-end sample sits in dBuf


@pytest.mark.gpu
def test_first_divergence_on_the_hip_engine():
    r = _run("--engine", "hip", "--blocks", "1100", "--seed", "5", "--profile", "mixed")
    assert r.returncode == 0 and "no divergence in 1100 blocks" in r.stdout, r.stdout + r.stderr
    r = _run("--engine", "hip", "--blocks", "400", "--seed", "5", "--perturb-block", "333", "--variant", "safe")
    assert r.returncode == 1 and "FIRST DIVERGENCE at block 333 " in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_soak_parity_tool_on_a_small_batch():
    """tools/soak_parity.py (every stream of a batch against the CPU checker) at a size that takes seconds: both input
    forms, and its verdict line must say what was compared."""
    import json
    tool = ROOT / "tools" / "soak_parity.py"
    for extra in ([], ["--clean", "--fs", "8000", "--variant", "safe"]):
        r = subprocess.run([sys.executable, str(tool), "--streams", "96", "--blocks", "160", "--passes", "2", "--chunk", "64", *extra],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["ok"] and line["streams"] == 96 and line["samples_compared"] == 96 * 160 * 64 and line["digests_compared"] == 96


def test_cli_wav_reader_equals_the_reference_clis_reader(tmp_path):
    """aecm_run reads every sample format the reference CLI reads through dr_wav (8 / 16 / 24 / 32-bit PCM, 32 / 64-bit
    float, A-law, mu-law; plain and WAVE_FORMAT_EXTENSIBLE headers, chunks to skip) and converts it to int16 by the same
    rule: `aecm_run --decode` (the reader on its own, no GPU) against oracle/_ref/ref_wavdec = dr_wav's
    drwav_open_file_and_read_pcm_frames_s16 (reference main.cc:39-54) behind a 20-line driver, sample for sample."""
    import wave
    from helpers import WAV_FORMATS, write_wav_format
    from oracle import pyoracle
    from webrtc_aecm_amd import build
    build.build()
    pyoracle.build()
    if not pyoracle.REFWAV.exists():
        pytest.skip("oracle/_ref/ref_wavdec not present (built from the reference's dr_wav.h where the reference tree exists)")
    rs = np.random.RandomState(5)
    x = np.concatenate([np.array([-32768, -32767, -1, 0, 1, 32766, 32767], dtype=np.int64), rs.randint(-32768, 32768, size=4000)])
    for fmt in WAV_FORMATS:
        src, ours, theirs = tmp_path / f"{fmt}.wav", tmp_path / f"{fmt}_ours.wav", tmp_path / f"{fmt}_ref.raw"
        write_wav_format(src, 16000, x, fmt)
        r = subprocess.run([str(pyoracle.REFWAV), str(src), str(theirs)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.split()[:2] == ["1", "16000"], (fmt, r.stdout, r.stderr)
        ref = np.fromfile(theirs, dtype="<i2")
        r = subprocess.run([str(build.CLI), "--decode", str(src), str(ours)], capture_output=True, text=True)
        assert r.returncode == 0, (fmt, r.stdout, r.stderr)
        with wave.open(str(ours), "rb") as w:
            assert w.getframerate() == 16000 and w.getnchannels() == 1 and w.getsampwidth() == 2
            got = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
        assert got.size == ref.size == x.size and np.array_equal(got, ref), (fmt, int(np.count_nonzero(got != ref)))
    # what dr_wav does not convert either: refused, not guessed
    bad = tmp_path / "f16.wav"
    import struct
    data = b"\0" * 64
    hdr = struct.pack("<HHIIHH", 3, 1, 16000, 32000, 2, 16)
    bad.write_bytes(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<I", 16) + hdr + b"data" + struct.pack("<I", len(data)) + data)
    assert subprocess.run([str(build.CLI), "--decode", str(bad), str(tmp_path / "o.wav")], capture_output=True).returncode != 0
