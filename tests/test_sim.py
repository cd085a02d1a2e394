"""CPU tests of the PRODUCT sources without a GPU:
  * webrtc_aecm_amd/csrc/aecm_wave.h (the block DSP the HIP kernel instantiates) on a 64-lane CPU
    simulator (tests/sim) against the oracle -- outputs and complete state, bit for bit;
  * webrtc_aecm_amd/csrc/aecm_session.cpp (host logic of the session ABI) over that simulator against
    the reference-generated session fixtures and, when present, the reference session ABI itself.
"""
import numpy as np
import pytest

import simlib
from helpers import adversarial_cases, describe_digest_diff, golden_files, run_burst_fixture
from oracle import pyoracle
from webrtc_aecm_amd.synth import synth_clean, synth_pair


@pytest.mark.parametrize("fs", [16000, 8000])
def test_wave_dsp_equals_oracle(fs):
    for seed in range(6):
        far, near = synth_pair(seed, 1400, fs)
        cng, em = (1 if seed % 7 else 0), seed % 5
        o, s = pyoracle.OracleStream(fs, cng, em), simlib.SimStream(fs, cng, em)
        for c in range(0, 1400, 350):        # also exercises state store/load between "launches"
            a = o.process(far[c * 64:(c + 350) * 64], near[c * 64:(c + 350) * 64])
            b = s.process(far[c * 64:(c + 350) * 64], near[c * 64:(c + 350) * 64])
            assert np.array_equal(a, b), (seed, c)
            assert np.array_equal(o.digest(), s.digest()), (seed, c, describe_digest_diff(o.digest(), s.digest()))


@pytest.mark.parametrize("order", [0, 1])
def test_role_decomposition_of_the_pipelined_kernel_equals_oracle(order):
    """The pipelined kernel's deepest shape splits a block over five roles with registers and state of their own -- front,
    delay, channel, gain, tail -- one step apart (aecm_block_kernels.hip).  The same split on the lane simulator
    (sim_process_roles: the same BlockEngine functions, state ownership, slot rings and far-history rules), launches of 1 .. 5 and
    several hundred blocks, alternating with the plain engine, fixed delays included: outputs and the complete state against the
    oracle after every launch.  order 0 / 1: consumers first / producers first inside a step -- a role reading what the same step
    writes would make the two differ."""
    for seed, (fixed, nlp) in enumerate([(None, 1), (0, 1), (1, 1), (7, 0), (None, 1), (99, 1)]):
        fs = 8000 if seed == 4 else 16000
        far, near = synth_pair(40 + seed, 1200, fs)
        cng, em = (0 if seed == 3 else 1), seed % 5
        o, s = pyoracle.OracleStream(fs, cng, em), simlib.SimStream(fs, cng, em)
        if fixed is not None:
            o.control(fixed, nlp)
            s.control(fixed, nlp)
        pos = 0
        for i, t in enumerate((1, 2, 3, 4, 5, 300, 120, 6, 559, 200)):
            f, n = far[pos * 64:(pos + t) * 64], near[pos * 64:(pos + t) * 64]
            a = o.process(f, n)
            b = s.process(f, n) if i in (6, 9) else s.process_roles(f, n, order)
            assert np.array_equal(a, b), (seed, i, order)
            assert np.array_equal(o.digest(), s.digest()), (seed, i, order, describe_digest_diff(o.digest(), s.digest()))
            pos += t
        assert pos == 1200


def test_wave_fft128_equals_oracle_fft_on_arbitrary_complex_data():
    """The kernel's fft128 (every per-stage scaling path of the inverse transform, the real-input
    forward specialisation and the generic complex forward) against the oracle's transform, which
    test_oracle pins to the reference's WebRtcSpl_ComplexFFT/IFFT.  The whole-block tests only ever
    feed the inverse transform conjugate-symmetric spectra; this feeds it anything."""
    import ctypes as C
    from test_oracle import fft_fuzz_cases
    olib, slib = pyoracle.oracle_lib(), simlib.lib()
    scales = set()
    for re, im in fft_fuzz_cases():
        for variant in (0, 1, 2):
            ore, oim = re.copy(), (np.zeros_like(im) if variant == 0 else im.copy())
            scale = C.c_int(0)
            olib.aecm_oracle_fft128(ore, oim, 1 if variant == 2 else 0, C.byref(scale))
            sre, sim_ = re.copy(), im.copy()
            got_scale = slib.sim_fft128(sre, sim_, variant)
            assert np.array_equal(sre, ore), (variant,)
            if variant != 2:
                assert np.array_equal(sim_[:64], oim[:64]), (variant,)
            else:
                assert got_scale == scale.value
                scales.add(got_scale)
    assert min(scales) == 0 and max(scales) >= 7 and len(scales) >= 8


def test_wave_dsp_rare_branches():
    fs = 16000
    far, near = synth_pair(100, 4200, fs, "silent")
    o, s = pyoracle.OracleStream(fs, 1, 3), simlib.SimStream(fs, 1, 3)
    assert np.array_equal(o.process(far, near), s.process(far, near))
    assert np.array_equal(o.digest(), s.digest())
    far, near = synth_pair(5, 900, fs)
    o, s = pyoracle.OracleStream(fs, 1, 3), simlib.SimStream(fs, 1, 3)
    o.control(7, 0)
    s.control(7, 0)
    assert np.array_equal(o.process(far, near), s.process(far, near))
    far, near = synth_pair(4, 600, fs)
    clean = (near.astype(np.int32) * 3 // 4).astype(np.int16)
    o, s = pyoracle.OracleStream(fs, 1, 2), simlib.SimStream(fs, 1, 2)
    exp = np.concatenate([o.process_block_clean(far[b * 64:(b + 1) * 64], near[b * 64:(b + 1) * 64],
                                                clean[b * 64:(b + 1) * 64]) for b in range(600)])
    assert np.array_equal(s.process(far, near, clean), exp)
    assert np.array_equal(o.digest(), s.digest())


def test_wave_dsp_full_scale_and_zero_inputs():
    rs = np.random.RandomState(3)
    n = 700 * 64
    cases = [
        (np.zeros(n, np.int16), np.zeros(n, np.int16)),
        (np.full(n, -32768, np.int16), np.full(n, -32768, np.int16)),
        (np.where(rs.randint(0, 2, n) == 1, 32767, -32768).astype(np.int16), rs.randint(-32768, 32768, n).astype(np.int16)),
        (rs.randint(-32768, 32768, n).astype(np.int16), np.zeros(n, np.int16)),
        (rs.randint(-2, 3, n).astype(np.int16), rs.randint(-2, 3, n).astype(np.int16)),
    ]
    for i, (far, near) in enumerate(cases):
        o, s = pyoracle.OracleStream(16000, 1, 3), simlib.SimStream(16000, 1, 3)
        assert np.array_equal(o.process(far, near), s.process(far, near)), i
        assert np.array_equal(o.digest(), s.digest()), i


def test_wave_dsp_adversarial_inputs_and_echo_paths():
    """Hostile signals, random configurations and random full-range echo paths through the kernel source
    on the lane simulator: bit-exact against the oracle, and none of the checked preconditions
    (mul24 operand range, as_i16 narrowing) may fire."""
    for it, c in enumerate(adversarial_cases()):
        o, s = pyoracle.OracleStream(c["fs"], c["cng"], c["echo_mode"]), simlib.SimStream(c["fs"], c["cng"], c["echo_mode"])
        if c["path"] is not None:
            o.init_echo_path(c["path"])
            s.init_echo_path(c["path"])
        assert np.array_equal(o.process(c["far"], c["near"]), s.process(c["far"], c["near"])), it
        assert np.array_equal(o.digest(), s.digest()), (it, describe_digest_diff(o.digest(), s.digest()))


def test_echo_path_import_export():
    path = (np.arange(65) * 53 % 3000).astype(np.int16)
    o, s = pyoracle.OracleStream(8000, 1, 3), simlib.SimStream(8000, 1, 3)
    o.init_echo_path(path)
    s.init_echo_path(path)
    assert np.array_equal(s.echo_path(), path)
    far, near = synth_pair(2, 500, 8000)
    assert np.array_equal(o.process(far, near), s.process(far, near))
    assert np.array_equal(o.echo_path(), s.echo_path())


def _run(sess, far, near, frame, ms, clean=None):
    out = near.copy()
    codes = set()
    for i in range(near.size // frame):
        sl = slice(i * frame, (i + 1) * frame)
        assert sess.buffer_farend(far[sl]) == 0
        rc, o = sess.process(out[sl], None if clean is None else clean[sl], ms)
        codes.add(rc)
        out[sl] = o
    return out, codes


def test_session_host_logic_matches_reference_fixtures():
    files = golden_files("session_")
    assert len(files) >= 5 and any("_clean" in f.name for f in files)
    for f in files:
        g = np.load(f)
        fs, frame, ms = int(g["fs"]), int(g["frame"]), int(g["ms"])
        far, near = synth_pair(int(g["seed"]), int(g["n_blocks"]), fs, "mixed")
        n = (far.size // frame) * frame
        clean = synth_clean(near)[:n] if "clean" in g.files and int(g["clean"]) else None   # nearendClean fixtures
        s = simlib.SimSession()
        assert s.init(fs) == 0 and s.set_config(int(g["cng"]), int(g["echo_mode"])) == 0
        out, codes = _run(s, far[:n], near[:n], frame, ms, clean)
        assert sorted(codes) == g["codes"].tolist(), f.name
        assert np.array_equal(out, g["out"]), f.name


def test_session_60s_cli_shaped_run_matches_reference_hash():
    """the reference CLI's procedure (main.cc: 160-sample calls, cng on, echoMode 1,
    ms 40) on a 60 s 16 kHz pair; the reference's result is pinned by its SHA-256."""
    import hashlib
    from helpers import GOLDEN
    far, near = synth_pair(60, 15000, 16000, "mixed")
    s = simlib.SimSession()
    assert s.init(16000) == 0 and s.set_config(1, 1) == 0
    out, codes = _run(s, far, near, 160, 40)
    assert codes == {0}
    assert hashlib.sha256(out.tobytes()).hexdigest() == (GOLDEN / "session_60s_16k.sha256").read_text().strip()


def test_session_error_codes_and_ownership_rules():
    s = simlib.SimSession()
    z = np.zeros(160, dtype=np.int16)
    assert s.buffer_farend(z) == 12002 and s.process(z)[0] == 12002 and s.set_config(1, 3) == 12002
    assert s.init(44100) == 12004
    assert s.init(16000) == 0
    assert s.buffer_farend(z[:100]) == 12004 and s.process(z[:100])[0] == 12004
    assert s.set_config(2, 3) == 12004 and s.set_config(1, 5) == 12004 and s.set_config(0, 0) == 0
    assert s.process(z, None, -5)[0] == 12100 and s.process(z, None, 501)[0] == 12100
    assert s.init_echo_path(z[:64]) == 12004
    rc, p = s.get_echo_path()
    assert rc == 0 and p[0] == 2040 and p[64] == 3153       # kChannelStored16kHz ends


@pytest.mark.skipif(not pyoracle.have_reference(), reason="oracle/_ref/libaecm_ref.so not built")
@pytest.mark.skipif(not pyoracle.have_reference(), reason="oracle/_ref/libaecm_ref.so not built")
def test_session_host_logic_matches_reference_abi_persistent_delays_and_clean_input():
    # Sound-card delays held for the whole session, including the ones that saturate the 4000-sample
    # jitter buffer (250 / 500 ms: the reference then drops new far-end data and keeps re-reading old
    # content), with and without a nearendClean input.
    for fs, frame, ms, with_clean in ((16000, 160, 250, 0), (16000, 160, 500, 1), (8000, 80, 500, 0), (8000, 160, 0, 1),
                                      (16000, 80, 130, 1)):
        far, near = synth_pair(41, 1200, fs, "mixed")
        clean = synth_clean(near) if with_clean else None
        r = pyoracle.RefSession(fs, 1, 2)
        s = simlib.SimSession()
        assert s.init(fs) == 0 and s.set_config(1, 2) == 0
        ob = np.empty(frame, dtype=np.int16)
        for i in range(far.size // frame):
            sl = slice(i * frame, (i + 1) * frame)
            assert r.lib.WebRtcAecm_BufferFarend(r.h, far[sl].ctypes.data, frame) == s.buffer_farend(far[sl])
            cp = clean[sl].ctypes.data if with_clean else None
            rc = r.lib.WebRtcAecm_Process(r.h, near[sl].ctypes.data, cp, ob.ctypes.data, frame, ms)
            rc2, o2 = s.process(near[sl], clean[sl] if with_clean else None, ms)
            assert rc == rc2 and np.array_equal(ob, o2), (fs, frame, ms, i)


@pytest.mark.skipif(not pyoracle.have_reference(), reason="oracle/_ref/libaecm_ref.so not built")
def test_session_host_logic_matches_reference_abi_odd_call_patterns():
    # 80-sample calls at 16 kHz (start-up never ends, nBlocks10ms == 0), 160-sample calls at 8 kHz,
    # a jittering msInSndCardBuf and a far-end underrun (BufferFarend skipped now and then).
    rs = np.random.RandomState(5)
    for fs, frame in ((16000, 80), (8000, 160), (16000, 160), (8000, 80)):
        far, near = synth_pair(31, 1500, fs, "mixed")
        r = pyoracle.RefSession(fs, 1, 3)
        s = simlib.SimSession()
        assert s.init(fs) == 0 and s.set_config(1, 3) == 0
        ob = np.empty(frame, dtype=np.int16)
        for i in range(far.size // frame):
            sl = slice(i * frame, (i + 1) * frame)
            ms = int(40 + rs.randint(-12, 13)) if i % 50 else int(rs.choice([-3, 0, 600, 90]))
            if i % 97 != 96:
                assert r.lib.WebRtcAecm_BufferFarend(r.h, far[sl].ctypes.data, frame) == s.buffer_farend(far[sl])
            rc = r.lib.WebRtcAecm_Process(r.h, near[sl].ctypes.data, None, ob.ctypes.data, frame, ms)
            rc2, o2 = s.process(near[sl], None, ms)
            assert rc == rc2 and np.array_equal(ob, o2), (fs, frame, i)


@pytest.mark.skipif(not pyoracle.have_reference(), reason="oracle/_ref/libaecm_ref.so not built")
@pytest.mark.parametrize("fs,frame", [(16000, 160), (8000, 80), (16000, 80), (8000, 160)])
def test_session_host_logic_far_end_bursts_and_mid_session_reconfiguration(fs, frame):
    """What a jittery network and a live application do to ONE session, against the reference's own ABI call by call:
    k = 0, 1, 1, 1, 2, 3 WebRtcAecm_BufferFarend calls per WebRtcAecm_Process, a 30-frame burst every 50 calls (the
    4 000-sample jitter buffer overflows and truncates: reference ring_buffer.c:142-170, echo_control_mobile.cc:215-234),
    a dozen calls without any far frame, and in between WebRtcAecm_set_config (valid and refused), InitEchoPath /
    GetEchoPath and WebRtcAecm_Init at the other sampling rate and back."""
    from helpers import call_pattern, drive_session, far_frames_needed, reconfiguration_events
    n_calls = 3 * fs // frame if frame == 160 or fs == 8000 else 400
    ms_seq, far_calls = call_pattern(7 + fs // 8000 + frame, n_calls, bursts=True)
    far, _ = synth_pair(61, far_frames_needed(far_calls) * frame // 64 + 1, fs, "mixed")
    _, near = synth_pair(61, n_calls * frame // 64 + 1, fs, "mixed")
    events = reconfiguration_events(fs, n_calls)
    r = pyoracle.RefSession(fs, 1, 3)
    s = simlib.SimSession()
    assert s.init(fs) == 0 and s.set_config(1, 3) == 0
    exp, exp_codes = drive_session(r, far, near, frame, ms_seq, far_calls, events=events)
    got, codes = drive_session(s, far, near, frame, ms_seq, far_calls, events=events)
    assert np.array_equal(codes, exp_codes)
    assert np.array_equal(got, exp), int(np.nonzero(got != exp)[0][0]) // frame
    assert len(r.event_log) == 3 and all(np.array_equal(a, b) for a, b in zip(r.event_log, s.event_log))
    assert (exp != near[:exp.size]).any()             # the session did leave its start-up copy


def test_batched_recordings_schedule_matches_reference_fixtures():
    """The index-domain session schedule (aecm_schedule.cpp) + gather/scatter reproduces, for every
    stream of a batch, what an individual WebRtcAecm_* session produces."""
    for f in golden_files("session_"):
        g = np.load(f)
        fs, frame, ms = int(g["fs"]), int(g["frame"]), int(g["ms"])
        far, near = synth_pair(int(g["seed"]), int(g["n_blocks"]), fs, "mixed")
        n = (far.size // frame) * frame
        far2, near2 = synth_pair(int(g["seed"]) + 50, int(g["n_blocks"]), fs, "mixed")
        nears = np.stack([near[:n], near2[:n]])
        cleans = synth_clean(nears) if "clean" in g.files and int(g["clean"]) else None
        rc, out = simlib.sim_recordings(np.stack([far[:n], far2[:n]]), nears, fs, frame, int(g["cng"]), int(g["echo_mode"]), ms,
                                        cleans)
        assert [rc] == [c for c in g["codes"].tolist()] or (rc == 0 and g["codes"].tolist() == [0]), f.name
        assert np.array_equal(out[0], g["out"]), f.name
        s = simlib.SimSession()
        assert s.init(fs) == 0 and s.set_config(int(g["cng"]), int(g["echo_mode"])) == 0
        exp, _ = _run(s, far2[:n], near2[:n], frame, ms, None if cleans is None else cleans[1])
        assert np.array_equal(out[1], exp), f.name


def test_host_built_kernel_constants_equal_their_definitions():
    """The GPU kernels read lane constants and LDS tables from a blob built on the host
    (BuildKernelConstants); it must equal what aecm_wave.h / the simulator compute from the definitions."""
    blob, rows, tw = simlib.constants()
    n = simlib.N_LANE_CONST_ROWS * 64
    assert np.array_equal(blob[:n], rows)
    assert np.array_equal(blob[n:n + tw.size], tw)
    hann = blob[n + tw.size + 360:]
    assert hann.size == 68 and hann[0] == 0 and hann[64] == 16384


def test_session_jitter_goldens_on_the_host_session_logic():
    """sessjit_* fixtures (reference outputs for a jittering msInSndCardBuf + far-end underruns): the product's
    Session class over the simulated engine must reproduce them call by call."""
    from helpers import drive_session
    files = golden_files("sessjit_")
    assert len(files) >= 3
    for f in files:
        g = np.load(f)
        fs, frame = int(g["fs"]), int(g["frame"])
        far, near = synth_pair(int(g["seed"]), int(g["n_blocks"]), fs, "mixed")
        s = simlib.SimSession()
        assert s.init(fs) == 0 and s.set_config(int(g["cng"]), int(g["echo_mode"])) == 0
        out, codes = drive_session(s, far, near, frame, g["ms_seq"], g["far_present"])
        assert np.array_equal(codes, g["codes"]) and np.array_equal(out, g["out"]), f.name


def test_session_burst_goldens_on_the_host_session_logic():
    """sessburst_* fixtures (reference outputs for far-end bursts, jitter-buffer overflow and mid-session set_config /
    InitEchoPath / re-Init at the other rate): the product's Session class over the simulated engine reproduces them."""
    files = golden_files("sessburst_")
    assert len(files) >= 2
    for f in files:
        g = np.load(f)
        assert int(g["far_calls"].max()) == 30 and int(g["far_calls"].min()) == 0
        out, codes, paths = run_burst_fixture(simlib.SimSession(), g)
        assert np.array_equal(codes, g["codes"]) and np.array_equal(out, g["out"]) and np.array_equal(paths, g["paths"]), f.name


@pytest.mark.parametrize("fs", [16000, 8000])
def test_device_session_machinery_equals_the_generic_wrapper_on_sample_tags(fs):
    """aecm_flow_plan.h (the wrapper + frame adapter as position arithmetic, what aecm_tick_flow_kernel runs per session)
    against SessionFlow<T> on sample tags: every block's 64 far / near inputs and every output sample must have the
    same provenance, tick by tick -- constant, jittering, stepping and out-of-range msInSndCardBuf, far-end underruns,
    a saturated jitter buffer (16 kHz in 80-sample calls), mixed 80 / 160 / 2 x 80 call shapes, a replay frame that
    outlives its place in the far ring, far-end bursts (k = 0..3 and 30 / 60 / 255 WebRtcAecm_BufferFarend calls between
    two WebRtcAecm_Process calls, overflowing the jitter buffer), and position counters that wrap around 2^32 and 2^31
    during the run.  Every state passed through must also be one ImportSession's validator accepts."""
    assert simlib.lib().sim_flow_tolerance_check() == 0
    saw_blocks = saw_drops = saw_direct = saw_framed = saw_spills = saw_bursts = saw_burst_drops = 0
    for scenario in list(range(11)) + [12, 13, 14]:
        for seed, start in ((1, 0), (2, 0xfffff000), (3, 0x7ffff800), (4, 987654321)):
            tick, detail = simlib.flow_fuzz(seed + 10 * scenario, fs, 6000, scenario, start)
            assert tick == -1, (fs, scenario, seed, tick, detail)
            saw_blocks += detail[1]
            saw_drops += detail[3]
            saw_direct += detail[4]
            saw_framed += detail[2] - detail[4]
            saw_spills += detail[5]
            saw_bursts += detail[6]
            saw_burst_drops += detail[7]
    # both far-end paths of the tick kernel (fetch from the far ring / through the framed-far ring), a saturated jitter
    # buffer and replay frames that had to move to their rows were all exercised
    assert saw_blocks > 100000 and saw_direct > 50000 and saw_framed > 50000 and saw_spills > 50
    assert fs == 8000 or saw_drops > 1000
    # far-end bursts (WebRtcAecm_BufferFarend calls without a Process, scenarios 12-14), some of them into a full jitter buffer
    assert saw_bursts > 20000 and saw_burst_drops > 100000
    if fs == 16000:     # counters wrapped negative during an endless 80-sample start-up, then compared as size_t (scenario 11)
        tick, detail = simlib.flow_fuzz(77, fs, 33600, 11, 12345)
        assert tick == -1 and detail[2] > 300, (tick, detail)


def test_wrapped_startup_counters_follow_the_reference():
    """The reference compares its short start-up counters with a size_t product (echo_control_mobile.cc:320,330).  At 16 kHz
    with 80-sample calls nBlocks10ms is 0, neither limit can fire, and after 32 768 calls the counters are negative: the
    first 160-sample call then sees a huge unsigned product and leaves the buffer-size check.  The host session logic
    (SessionFlow, which the device plan is fuzzed against) must follow the real reference through exactly that."""
    if not pyoracle.have_reference():
        pytest.skip("needs oracle/_ref")
    fs, n80, n160 = 16000, 33000, 60
    far, near = synth_pair(4242, (n160 * 160) // 64 + 2, fs, "steady")
    z = np.zeros(80, np.int16)
    r = pyoracle.RefSession(fs, 1, 3)
    s = simlib.SimSession()
    assert s.init(fs) == 0 and s.set_config(1, 3) == 0
    ob = np.empty(80, dtype=np.int16)
    for i in range(n80):
        ms = 40 + i % 3
        if i % 400 == 0:      # the far end is offered now and then (the jitter buffer saturates either way)
            assert r.lib.WebRtcAecm_BufferFarend(r.h, z.ctypes.data, 80) == s.buffer_farend(z)
        rc = r.lib.WebRtcAecm_Process(r.h, z.ctypes.data, None, ob.ctypes.data, 80, ms)
        rc2, o2 = s.process(z, None, ms)
        assert rc == rc2 and np.array_equal(ob, o2), i
    ob = np.empty(160, dtype=np.int16)
    for i in range(n160):
        sl = slice(i * 160, (i + 1) * 160)
        assert r.lib.WebRtcAecm_BufferFarend(r.h, far[sl].ctypes.data, 160) == s.buffer_farend(far[sl])
        rc = r.lib.WebRtcAecm_Process(r.h, near[sl].ctypes.data, None, ob.ctypes.data, 160, 40)
        rc2, o2 = s.process(near[sl], None, 40)
        assert rc == rc2 and np.array_equal(ob, o2), i


def test_reciprocal_division_of_the_nlms_step_is_exact():
    """divu_by_magic with div_magic's 33-bit reciprocals == n / d for d = 1..65 (bin + 1) over [0, 2^31]."""
    assert simlib.lib().sim_div_magic_check() == 0


def test_inverse_fft_growth_bound_behind_the_joint_scaling_tests():
    """aecm_wave.h skips per-stage scaling tests of the inverse transform when max |x| at the group's first stage proves
    that no stage of the group can reach 13 573: an unscaled stage grows the largest magnitude M to less than
    M (32768 + L) / 32768 + 2 with L = 46342 >= |wr| + |wi| for every twiddle.  Checked here against an exact integer model
    of the unscaled stages (complex_fft.c:465-482 with shift 0) on sign patterns and on a greedy adversary that flips signs
    to maximise the growth -- and the resulting bounds must be the ones the kernel asserts (2 327 / 5 621)."""
    import math
    sin1024 = [int(32767.0 * math.sin(2.0 * math.pi * i / 1024.0)) for i in range(1024)]
    L = max(abs(sin1024[8 * r + 256]) + abs(sin1024[8 * r]) for r in range(64))
    assert L <= 46342

    def growth(m):
        return (m * (32768 + 46342) >> 15) + 2

    def bound(k):
        best = 0
        for m in range(1, 13574):
            v = m
            for _ in range(k - 1):
                v = growth(v)
            if v > 13573:
                break
            best = m
        return best
    assert (bound(1), bound(2), bound(3)) == (13573, 5621, 2327)

    def stage(re, im, st):                       # one unscaled inverse stage on bit-reversed-order data, in place (int64 arrays)
        l, k = 1 << st, 9 - st
        for m in range(l):
            r = (m << k) >> 3
            wr, wi = sin1024[8 * r + 256], sin1024[8 * r]
            i = np.arange(m, 128, 2 * l)
            j = i + l
            tr = (wr * re[j] - wi * im[j] + 1) >> 1
            ti = (wr * im[j] + wi * re[j] + 1) >> 1
            qr, qi = re[i] * 16384, im[i] * 16384
            re[j], im[j] = (qr - tr + 8192) >> 14, (qi - ti + 8192) >> 14
            re[i], im[i] = (qr + tr + 8192) >> 14, (qi + ti + 8192) >> 14

    def worst_after(re0, im0, first, n_stages):
        re, im = re0.astype(np.int64), im0.astype(np.int64)
        peaks = []
        for st in range(first, first + n_stages):
            m_in = int(max(np.abs(re).max(), np.abs(im).max()))
            stage(re, im, st)
            m_out = int(max(np.abs(re).max(), np.abs(im).max()))
            assert m_out <= growth(m_in), (st, m_in, m_out)
            peaks.append(m_out)
        return peaks
    rs = np.random.RandomState(3)
    for first, n_stages, amp in ((0, 3, 2327), (3, 2, 5621), (5, 2, 5621), (2, 3, 2327), (4, 3, 2327)):
        best = 0
        for trial in range(40):
            sr, si = rs.choice([-1, 1], 128), rs.choice([-1, 1], 128)
            peak = worst_after(amp * sr, amp * si, first, n_stages)[-2] if n_stages > 1 else amp
            # greedy adversary: flip one sign at a time while the magnitude entering the group's LAST stage grows
            for _ in range(60):
                k = rs.randint(0, 256)
                (sr if k < 128 else si)[k % 128] *= -1
                p2 = worst_after(amp * sr, amp * si, first, n_stages)[-2] if n_stages > 1 else amp
                if p2 >= peak:
                    peak = p2
                else:
                    (sr if k < 128 else si)[k % 128] *= -1
            best = max(best, peak)
        assert best <= 13573, (first, n_stages, amp, best)          # the group's last stage never sees a magnitude that scales
