"""CPU tests that pin the oracle: against the committed golden vectors (generated from the unmodified
reference by tools/gen_golden.py) and, where oracle/_ref/libaecm_ref.so exists, against the reference
itself on seeded and adversarial inputs."""
import hashlib

import numpy as np
import pytest

from helpers import adversarial_cases, describe_digest_diff, golden_files
from oracle import pyoracle
from webrtc_aecm_amd.synth import PROFILES, synth_pair

needs_ref = pytest.mark.skipif(not pyoracle.have_reference(), reason="oracle/_ref/libaecm_ref.so not built")


def test_synth_is_deterministic():
    a = synth_pair(3, 200, 16000)
    b = synth_pair(3, 200, 16000)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # pinned so that a numpy/generator change on another machine cannot silently move the fixtures
    h = hashlib.sha256(synth_pair(0, 100, 16000)[1].tobytes()).hexdigest()
    assert h == hashlib.sha256(synth_pair(0, 100, 16000, PROFILES[0])[1].tobytes()).hexdigest()


def test_oracle_matches_golden_block_vectors():
    files = golden_files("block_")
    assert len(files) >= 5
    for f in files:
        g = np.load(f)
        seed, nb, fs = int(g["seed"]), int(g["n_blocks"]), int(g["fs"])
        prof = str(g["profile"]) or None
        far, near = synth_pair(seed, nb, fs, prof)
        o = pyoracle.OracleStream(fs, int(g["cng"]), int(g["echo_mode"]))
        outs = []
        for i, c in enumerate(range(0, nb, 300)):
            outs.append(o.process(far[c * 64:(c + 300) * 64], near[c * 64:(c + 300) * 64]))
            d = o.digest()
            assert np.array_equal(d, g["digests"][i]), f"{f.name} block {c + 300}: {describe_digest_diff(d, g['digests'][i])}"
        out = np.concatenate(outs)
        assert hashlib.sha256(out.tobytes()).hexdigest() == str(g["sha256"]), f.name
        assert np.array_equal(out[-g["out"].size:], g["out"])


def test_sqrt_floor_is_exact_floor_sqrt():
    lib = pyoracle.oracle_lib()
    rs = np.random.RandomState(1)
    vals = np.concatenate([np.arange(0, 70000), rs.randint(0, 2 ** 31 - 1, size=200000),
                           np.array([2 ** 31 - 1, 46340 ** 2, 46340 ** 2 - 1, 46340 ** 2 + 1])])
    for v in vals[::7]:
        r = lib.aecm_oracle_sqrt_floor(int(v))
        assert r * r <= v < (r + 1) * (r + 1)


def test_forward_fft_of_impulse_and_dc():
    lib = pyoracle.oracle_lib()
    re = np.zeros(128, dtype=np.int16)
    im = np.zeros(128, dtype=np.int16)
    re[0] = 12800
    lib.aecm_oracle_fft128(re, im, 0, None)
    assert np.all(re == 100) and np.all(im == 0)          # impulse / 128
    re[:] = 6400
    im[:] = 0
    lib.aecm_oracle_fft128(re, im, 0, None)
    assert re[0] == 6400 and np.all(re[1:] == 0) and np.all(im == 0)


def fft_fuzz_cases(n_cases=400, seed=5):
    """(re, im) int16 pairs covering every per-stage scaling regime of the inverse transform:
    amplitudes from a few LSB to full scale, plus the -32768 / 32767 corner patterns."""
    rng = np.random.default_rng(seed)
    cases = []
    for k in range(n_cases):
        amp = int(2 ** rng.uniform(1, 15.2))
        re = rng.integers(-min(amp, 32768), min(amp, 32767) + 1, 128).astype(np.int16)
        im = rng.integers(-min(amp, 32768), min(amp, 32767) + 1, 128).astype(np.int16)
        if k % 7 == 0:
            re[rng.integers(0, 128, 4)] = -32768
        if k % 11 == 0:
            im[rng.integers(0, 128, 4)] = 32767
        if k % 13 == 0:
            re[:] = np.where(np.arange(128) % 2 == 0, 32767, -32768)
        cases.append((re, im))
    cases.append((np.full(128, -32768, np.int16), np.full(128, -32768, np.int16)))
    cases.append((np.zeros(128, np.int16), np.zeros(128, np.int16)))
    # amplitudes at and just above the bounds the kernel's joint scaling tests use (aecm_wave.h: no_scale_bound<3>, <2>,
    # the reference's two thresholds), as +-amplitude sign patterns -- the inputs that grow fastest through the stages
    n = np.arange(128)
    for b in (2327, 5621, 13573, 27146):
        for amp in (b - 1, b, b + 1):
            pats = [(np.ones(128), np.ones(128)), (1 - 2 * (n % 2), np.ones(128)), (1 - 2 * ((n // 2) % 2), 1 - 2 * (n % 2)),
                    (1 - 2 * ((n // 4) % 2), 1 - 2 * ((n // 8) % 2))]
            pats += [(rng.choice([-1, 1], 128), rng.choice([-1, 1], 128)) for _ in range(6)]
            for sr, si in pats:
                cases.append(((amp * sr).astype(np.int16), (amp * si).astype(np.int16)))
            sparse = np.zeros(128, np.int16)
            sparse[rng.integers(0, 128)] = amp                       # one element at the bound, the rest silent
            cases.append((sparse, np.zeros(128, np.int16)))
    return cases


@needs_ref
def test_oracle_fft_equals_reference_complex_fft():
    """aecm_oracle_fft128 (bit reversal + transform) against the reference's own
    WebRtcSpl_ComplexBitReverse + WebRtcSpl_ComplexFFT / ComplexIFFT (complex_fft.c:181-491)."""
    import ctypes as C
    olib, rlib = pyoracle.oracle_lib(), pyoracle.ref_lib()
    i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
    rlib.WebRtcSpl_ComplexBitReverse.argtypes = [i16p, C.c_int]
    rlib.WebRtcSpl_ComplexFFT.argtypes = [i16p, C.c_int, C.c_int]
    rlib.WebRtcSpl_ComplexIFFT.argtypes = [i16p, C.c_int, C.c_int]
    shifts_seen = set()
    for re, im in fft_fuzz_cases():
        for inverse in (0, 1):
            frfi = np.empty(256, dtype=np.int16)
            frfi[0::2], frfi[1::2] = re, im
            rlib.WebRtcSpl_ComplexBitReverse(frfi, 7)
            ref_scale = (rlib.WebRtcSpl_ComplexIFFT if inverse else rlib.WebRtcSpl_ComplexFFT)(frfi, 7, 1)
            ore, oim = re.copy(), im.copy()
            scale = C.c_int(0)
            olib.aecm_oracle_fft128(ore, oim, inverse, C.byref(scale))
            assert np.array_equal(ore, frfi[0::2]) and np.array_equal(oim, frfi[1::2])
            if inverse:
                assert scale.value == ref_scale
                shifts_seen.add(scale.value)
    assert min(shifts_seen) == 0 and len(shifts_seen) >= 8       # from "never scales" to "scales at every stage"


@needs_ref
@pytest.mark.parametrize("fs", [16000, 8000])
def test_oracle_equals_reference_on_seeded_streams(fs):
    for seed in range(8):
        far, near = synth_pair(seed, 2048, fs)
        cng, em = (1 if seed % 7 else 0), seed % 5
        o, r = pyoracle.OracleStream(fs, cng, em), pyoracle.RefCoreStream(fs, cng, em)
        for c in range(0, 2048, 256):
            a = o.process(far[c * 64:(c + 256) * 64], near[c * 64:(c + 256) * 64])
            b = r.process(far[c * 64:(c + 256) * 64], near[c * 64:(c + 256) * 64])
            assert np.array_equal(a, b), (seed, c)
            assert np.array_equal(o.digest(), r.digest()), (seed, c, describe_digest_diff(o.digest(), r.digest()))


@needs_ref
def test_oracle_equals_reference_long_silence_control_and_clean():
    for fs in (16000, 8000):
        far, near = synth_pair(100, 5000, fs, "silent")         # noise-floor floor branches
        o, r = pyoracle.OracleStream(fs, 1, 3), pyoracle.RefCoreStream(fs, 1, 3)
        assert np.array_equal(o.process(far, near), r.process(far, near))
        assert np.array_equal(o.digest(), r.digest())
        far, near = synth_pair(5, 1200, fs)                     # WebRtcAecm_Control: fixed delay, NLP off
        o, r = pyoracle.OracleStream(fs, 1, 3), pyoracle.RefCoreStream(fs, 1, 3)
        o.control(7, 0)
        r.control(7, 0)
        assert np.array_equal(o.process(far, near), r.process(far, near))
        far, near = synth_pair(4, 700, fs)                      # nearendClean path
        clean = (near.astype(np.int32) * 3 // 4).astype(np.int16)
        o, r = pyoracle.OracleStream(fs, 1, 2), pyoracle.RefCoreStream(fs, 1, 2)
        for b in range(700):
            sl = slice(b * 64, (b + 1) * 64)
            assert np.array_equal(o.process_block_clean(far[sl], near[sl], clean[sl]),
                                  r.process_block_clean(far[sl], near[sl], clean[sl])), b


@needs_ref
def test_oracle_equals_reference_on_adversarial_inputs():
    """Hostile signals, random configurations and random full-range echo paths (WebRtcAecm_InitEchoPath)."""
    for it, c in enumerate(adversarial_cases()):
        o, r = pyoracle.OracleStream(c["fs"], c["cng"], c["echo_mode"]), pyoracle.RefCoreStream(c["fs"], c["cng"], c["echo_mode"])
        if c["path"] is not None:
            o.init_echo_path(c["path"])
            r.init_echo_path(c["path"])
        assert np.array_equal(o.process(c["far"], c["near"]), r.process(c["far"], c["near"])), it
        assert np.array_equal(o.digest(), r.digest()), it
