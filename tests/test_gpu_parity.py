"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, the golden vectors
generated from the unmodified reference, and -- when the prebuilt oracle/_ref travels with the
snapshot -- the reference itself.  Bit-exact: every comparison is array_equal on int16 / uint32.
"""
import hashlib
import os

import numpy as np
import pytest

import webrtc_aecm_amd as aecm
from helpers import (GOLDEN, adversarial_cases, call_pattern, describe_digest_diff, drive_session, far_frames_needed, golden_files,
                     oracle_batch, oracle_run, reconfiguration_events, run_burst_fixture, stream_config, synth_streams)
from oracle import pyoracle
from webrtc_aecm_amd.synth import synth_clean, synth_pair

pytestmark = pytest.mark.gpu


def test_wave_primitives_selftest():
    """DPP / permlane / readlane primitives == their ds_bpermute definitions; floor-sqrt exhaustive."""
    f = aecm.self_test(0, exhaustive=True)
    assert f.tolist() == [0] * 8, f"self-test failures {f.tolist()} (see csrc/aecm_kernels.h for the index meaning)"


@pytest.mark.parametrize("variant", [aecm.KERNEL_FAST, aecm.KERNEL_SAFE])
def test_device_fft128_on_arbitrary_complex_data(variant):
    """The kernel's own transform code on the device (every per-stage scaling path of the inverse,
    the multiply-add forward stages, the generic complex forward) against the oracle's transform,
    which test_oracle pins to the reference's WebRtcSpl_ComplexFFT / ComplexIFFT."""
    import ctypes as C
    from test_oracle import fft_fuzz_cases
    olib = pyoracle.oracle_lib()
    cases = fft_fuzz_cases(n_cases=600, seed=11)
    re = np.stack([c[0] for c in cases])
    im = np.stack([c[1] for c in cases])
    for fft_variant in (0, 1, 2):
        gre, gim, gscale = aecm.debug_fft128(re, im, fft_variant, variant)
        for k in range(re.shape[0]):
            ore, oim = re[k].copy(), (np.zeros(128, np.int16) if fft_variant == 0 else im[k].copy())
            scale = C.c_int(0)
            olib.aecm_oracle_fft128(ore, oim, 1 if fft_variant == 2 else 0, C.byref(scale))
            assert np.array_equal(gre[k], ore), (fft_variant, k)
            if fft_variant != 2:
                assert np.array_equal(gim[k][:64], oim[:64]), (fft_variant, k)
            else:
                assert gscale[k] == scale.value, (fft_variant, k)


@pytest.mark.parametrize("variant,pipelined", [(aecm.KERNEL_SAFE, False), (aecm.KERNEL_FAST, False), (aecm.KERNEL_FAST, True)])
@pytest.mark.parametrize("fs", [16000, 8000])
def test_block_parity_vs_oracle(fs, variant, pipelined):
    """64 streams x 2048 blocks (all three start-up states), mixed configs, outputs and full state; the fast variant in
    both forms a launch of this size can take: one wavefront per stream, and pipelined (six wavefronts per four streams)."""
    S, T = 64, 2048
    seeds = list(range(1000, 1000 + S))
    cfgs = [stream_config(s) for s in range(S)]
    far, near = synth_streams(seeds, T, fs)
    b = aecm.AecmBatch(S, fs, variant=variant)
    b.set_launch_pipelining(2 if pipelined else 0)
    assert b.describe_launch(T)[0] == (3 if pipelined else 0)
    for s, (cng, em) in enumerate(cfgs):
        b.set_config(cng, em, s, 1)
    out = b.process_host(far, near)
    exp_out, exp_dig = oracle_batch(seeds, T, fs, cfgs, pairs=(far, near))
    bad = [s for s in range(S) if not np.array_equal(out[s], exp_out[s])]
    assert not bad, f"output mismatch in streams {bad[:8]} (first bad sample of stream {bad[0]}: " \
                    f"{int(np.nonzero(out[bad[0]] != exp_out[bad[0]])[0][0])})"
    for s in range(S):
        d = b.digest(s)
        assert np.array_equal(d, exp_dig[s]), f"state digest mismatch stream {s}: {describe_digest_diff(d, exp_dig[s])}"


@pytest.mark.parametrize("variant", [aecm.KERNEL_FAST, aecm.KERNEL_SAFE])
def test_adversarial_inputs_and_echo_paths(variant):
    """Hostile signals (full-scale noise / square waves / -32768 constants / spikes / LSB dither / tones),
    random configurations and random full-range echo paths, one case per stream of a batch, both
    sampling rates: outputs and complete state against the oracle."""
    cases = list(adversarial_cases())
    for fs in (16000, 8000):
        sel = [c for c in cases if c["fs"] == fs]
        b = aecm.AecmBatch(len(sel), fs, variant=variant)
        for k, c in enumerate(sel):
            b.set_config(c["cng"], c["echo_mode"], k, 1)
            if c["path"] is not None:
                b.init_echo_path(k, c["path"])
        out = b.process_host(np.stack([c["far"] for c in sel]), np.stack([c["near"] for c in sel]))
        for k, c in enumerate(sel):
            o = pyoracle.OracleStream(fs, c["cng"], c["echo_mode"])
            if c["path"] is not None:
                o.init_echo_path(c["path"])
            assert np.array_equal(out[k], o.process(c["far"], c["near"])), (fs, k)
            d = b.digest(k)
            assert np.array_equal(d, o.digest()), (fs, k, describe_digest_diff(d, o.digest()))


def test_host_buffer_pipeline_many_streams():
    """WebRtcAecmBatch_ProcessBlocksHost with enough streams for the chunked upload / compute / download
    pipeline (ragged last chunk), with and without the clean input: every stream against the oracle."""
    S, T, K = 2 * 8192 + 260, 40, 16
    pairs = [synth_pair(300 + k, T, 16000) for k in range(K)]
    idx = np.arange(S) % K
    far = np.stack([p[0] for p in pairs])[idx]
    near = np.stack([p[1] for p in pairs])[idx]
    for with_clean in (False, True):
        clean = synth_clean(near) if with_clean else None
        b = aecm.AecmBatch(S, 16000, 1, 3)
        out = b.process_host(far, near, clean)
        for k in range(K):
            o = pyoracle.OracleStream(16000, 1, 3)
            exp = o.process(pairs[k][0], pairs[k][1]) if not with_clean else np.concatenate(
                [o.process_block_clean(pairs[k][0][j * 64:(j + 1) * 64], pairs[k][1][j * 64:(j + 1) * 64],
                                       synth_clean(pairs[k][1])[j * 64:(j + 1) * 64]) for j in range(T)])
            rows = out[idx == k]
            assert np.array_equal(rows, np.broadcast_to(exp, rows.shape)), (with_clean, k)
        for s in (0, 8191, 8192, S - 1):
            o = pyoracle.OracleStream(16000, 1, 3)
            if with_clean:
                for j in range(T):
                    o.process_block_clean(far[s][j * 64:(j + 1) * 64], near[s][j * 64:(j + 1) * 64], clean[s][j * 64:(j + 1) * 64])
            else:
                o.process(far[s], near[s])
            assert np.array_equal(b.digest(s), o.digest()), (with_clean, s)
        b.close()


def test_chunked_launches_equal_one_launch():
    """State written back at the end of a launch and reloaded by the next must lose nothing."""
    S, T, fs = 16, 1300, 16000
    seeds = list(range(50, 50 + S))
    far, near = synth_streams(seeds, T, fs)
    one = aecm.AecmBatch(S, fs)
    ref = one.process_host(far, near)
    chunked = aecm.AecmBatch(S, fs)
    outs, pos = [], 0
    for n in [1, 1, 2, 3, 5, 64, 100, 7, 511, 1, 605]:
        outs.append(chunked.process_host(far[:, pos * 64:(pos + n) * 64], near[:, pos * 64:(pos + n) * 64]))
        pos += n
    assert pos == T
    assert np.array_equal(np.concatenate(outs, axis=1), ref)
    for s in range(S):
        assert np.array_equal(one.digest(s), chunked.digest(s))


def test_ragged_batch_shapes_and_empty_calls():
    """Stream counts that do not fill a workgroup, one-block launches, zero-block calls, bad arguments."""
    fs = 16000
    lib = aecm.load()
    for S in (1, 3, 5, 67):
        seeds = list(range(800, 800 + S))
        far, near = synth_streams(seeds, 70, fs)
        b = aecm.AecmBatch(S, fs)
        outs = []
        pos = 0
        for n in (1, 1, 2, 63, 3):
            outs.append(b.process_host(far[:, pos * 64:(pos + n) * 64], near[:, pos * 64:(pos + n) * 64]))
            pos += n
        got = np.concatenate(outs, axis=1)
        for s in range(S):
            exp, dig = oracle_run(seeds[s], 70, fs, 1, 3)
            assert np.array_equal(got[s], exp), (S, s)
            assert np.array_equal(b.digest(s), dig), (S, s)
        z = np.zeros((S, 0), dtype=np.int16)
        assert lib.WebRtcAecmBatch_ProcessBlocksHost(b.h, far.ctypes.data, near.ctypes.data, None, got.ctypes.data, 0, 64, 0) == 0
        assert lib.WebRtcAecmBatch_ProcessBlocksHost(b.h, far.ctypes.data, near.ctypes.data, None, got.ctypes.data, 0, 64, -1) == 12004
        assert lib.WebRtcAecmBatch_ProcessBlocksHost(b.h, None, near.ctypes.data, None, got.ctypes.data, 64, 64, 1) == 12003
        assert lib.WebRtcAecmBatch_GetDigest(b.h, S, np.zeros(24, np.uint32).ctypes.data) == 12004
        assert lib.WebRtcAecmBatch_set_config(b.h, aecm.AecmConfig(1, 9), 0, -1) == 12004
    assert lib.WebRtcAecmBatch_Init(None, 16000) == -1
    assert lib.WebRtcAecmBatch_Create(0, 0) is None
    assert lib.WebRtcAecmBatch_Create(4, 99) is None            # no such device


def test_golden_block_vectors():
    """Committed outputs + digests produced by the unmodified reference (tools/gen_golden.py)."""
    files = golden_files("block_")
    assert files
    for f in files:
        g = np.load(f)
        seed, nb, fs = int(g["seed"]), int(g["n_blocks"]), int(g["fs"])
        prof = str(g["profile"]) or None
        far, near = synth_pair(seed, nb, fs, prof)
        b = aecm.AecmBatch(1, fs, int(g["cng"]), int(g["echo_mode"]))
        outs = []
        for i, c in enumerate(range(0, nb, 300)):
            outs.append(b.process_host(far[None, c * 64:(c + 300) * 64], near[None, c * 64:(c + 300) * 64])[0])
            assert np.array_equal(b.digest(0), g["digests"][i]), f"{f.name}: digest after block {c + 300}"
        out = np.concatenate(outs)
        assert hashlib.sha256(out.tobytes()).hexdigest() == str(g["sha256"]), f.name
        assert np.array_equal(out[-g["out"].size:], g["out"]), f.name


def test_clean_input_path():
    """The optional nearendClean input (third transform, reference aecm_core_c.cc:449-464)."""
    S, T, fs = 8, 900, 16000
    seeds = list(range(300, 300 + S))
    far, near = synth_streams(seeds, T, fs)
    clean = (near.astype(np.int32) * 3 // 4).astype(np.int16)
    b = aecm.AecmBatch(S, fs)
    out = b.process_host(far, near, clean)
    for s in range(S):
        o = pyoracle.OracleStream(fs, 1, 3)
        exp = np.concatenate([o.process_block_clean(far[s, k * 64:(k + 1) * 64], near[s, k * 64:(k + 1) * 64],
                                                    clean[s, k * 64:(k + 1) * 64]) for k in range(T)])
        assert np.array_equal(out[s], exp), f"stream {s}"
        assert np.array_equal(b.digest(s), o.digest())


def test_control_fixed_delay_and_nlp_off():
    S, T, fs = 4, 800, 16000
    seeds = [5, 6, 7, 8]
    far, near = synth_streams(seeds, T, fs)
    b = aecm.AecmBatch(S, fs)
    b.control(7, 0)
    out = b.process_host(far, near)
    for s in range(S):
        o = pyoracle.OracleStream(fs, 1, 3)
        o.control(7, 0)
        assert np.array_equal(out[s], o.process(far[s], near[s]))


def test_tick_major_layout_and_device_pointers():
    """[T][S][64] device buffers through WebRtcAecmBatch_ProcessBlocks (strides 64, S*64)."""
    import torch
    S, T, fs = 32, 600, 16000
    seeds = list(range(700, 700 + S))
    far, near = synth_streams(seeds, T, fs)
    exp = aecm.AecmBatch(S, fs).process_host(far, near)
    tm = lambda a: torch.from_numpy(np.ascontiguousarray(a.reshape(S, T, 64).transpose(1, 0, 2))).cuda()
    dfar, dnear = tm(far), tm(near)
    dout = torch.zeros_like(dnear)
    torch.cuda.synchronize()
    b = aecm.AecmBatch(S, fs)
    b.process_device(dfar.data_ptr(), dnear.data_ptr(), dout.data_ptr(), 64, S * 64, T)
    b.synchronize()
    got = dout.cpu().numpy().transpose(1, 0, 2).reshape(S, T * 64)
    assert np.array_equal(got, exp)
    assert b.last_launch_ms() > 0


def test_echo_path_roundtrip_and_reinit():
    b = aecm.AecmBatch(3, 16000)
    path = (np.arange(65) * 37 % 4000).astype(np.int16)
    b.init_echo_path(1, path)
    assert np.array_equal(b.get_echo_path(1), path)
    o = pyoracle.OracleStream(16000, 1, 3)
    o.init_echo_path(path)
    far, near = synth_pair(21, 400, 16000)
    fb = np.stack([far] * 3)
    nb = np.stack([near] * 3)
    out = b.process_host(fb, nb)
    assert np.array_equal(out[1], o.process(far, near))
    assert np.array_equal(b.get_echo_path(1), o.echo_path())
    assert not np.array_equal(out[0], out[1])


def test_state_snapshot_migrates_a_stream():
    """Export a stream's state mid-run, import it into another batch (other slot, other config): both
    continue bit-exactly."""
    fs, T1, T2 = 16000, 700, 500
    far, near = synth_streams([61, 62, 63], T1 + T2, fs)
    a = aecm.AecmBatch(3, fs, 1, 2)
    a.process_host(far[:, :T1 * 64], near[:, :T1 * 64])
    blob = a.export_state(2)
    assert len(blob) == aecm.load().WebRtcAecmBatch_state_size_bytes() == 32 + 12 * 256 + 256 + 100 * 128   # header + vec + scal + hist
    b = aecm.AecmBatch(2, 8000, 0, 4)                      # deliberately different rate/config: all of it is state
    b.import_state(1, blob)
    out_a = a.process_host(far[:, T1 * 64:], near[:, T1 * 64:])
    fb = np.stack([far[0, T1 * 64:], far[2, T1 * 64:]])
    nb = np.stack([near[0, T1 * 64:], near[2, T1 * 64:]])
    out_b = b.process_host(fb, nb)
    assert np.array_equal(out_b[1], out_a[2])
    assert np.array_equal(b.digest(1), a.digest(2))
    exp, dig = oracle_run(63, T1 + T2, fs, 1, 2)
    assert np.array_equal(out_a[2], exp[T1 * 64:]) and np.array_equal(a.digest(2), dig)


def _run_session(sess, far, near, frame, ms, clean=None):
    out = near.copy()
    codes = set()
    for i in range(near.size // frame):
        sl = slice(i * frame, (i + 1) * frame)
        assert sess.buffer_farend(far[sl]) == 0
        rc, o = sess.process(out[sl], None if clean is None else clean[sl], ms)
        codes.add(rc)
        out[sl] = o
    return out, codes


def _fixture_has_clean(g):
    return "clean" in g.files and int(g["clean"]) != 0


def test_session_abi_golden():
    """The drop-in WebRtcAecm_* session ABI on the GPU against reference-generated fixtures
    (including WebRtcAecm_Process with a nearendClean input)."""
    files = golden_files("session_")
    assert files and any("_clean" in f.name for f in files)
    for f in files:
        g = np.load(f)
        fs, frame, ms = int(g["fs"]), int(g["frame"]), int(g["ms"])
        far, near = synth_pair(int(g["seed"]), int(g["n_blocks"]), fs, "mixed")
        n = (far.size // frame) * frame
        clean = synth_clean(near)[:n] if _fixture_has_clean(g) else None
        s = aecm.Aecm()
        assert s.init(fs) == 0
        assert s.set_config(int(g["cng"]), int(g["echo_mode"])) == 0
        out, codes = _run_session(s, far[:n], near[:n], frame, ms, clean)
        assert sorted(codes) == g["codes"].tolist(), f.name
        assert np.array_equal(out, g["out"]), f.name
        s.close()


def test_session_60s_cli_shaped_run_matches_reference_hash():
    """one stream, 60 s at 16 kHz, through the session ABI exactly as the reference
    CLI drives it (main.cc:102-145); the reference's output is pinned by its SHA-256.  Also as one
    recording of a device batch."""
    far, near = synth_pair(60, 15000, 16000, "mixed")
    want = (GOLDEN / "session_60s_16k.sha256").read_text().strip()
    s = aecm.Aecm()
    assert s.init(16000) == 0 and s.set_config(1, 1) == 0
    out, codes = _run_session(s, far, near, 160, 40)
    s.close()
    assert codes == {0} and hashlib.sha256(out.tobytes()).hexdigest() == want
    b = aecm.AecmBatch(2, 16000, 1, 1)
    rc, outs = b.process_recordings_host(np.stack([far, far]), np.stack([near, near]), 160, 40)
    assert rc == 0 and hashlib.sha256(outs[1].tobytes()).hexdigest() == want


def test_batched_recordings_equal_individual_sessions():
    """WebRtcAecmBatch_ProcessRecordingsHost: S recordings as S sessions in one device batch (with and
    without a nearendClean input)."""
    for f in golden_files("session_"):
        g = np.load(f)
        fs, frame, ms = int(g["fs"]), int(g["frame"]), int(g["ms"])
        S = 5
        pairs = [synth_pair(int(g["seed"]) + 50 * k, int(g["n_blocks"]), fs, "mixed") for k in range(S)]
        n = (pairs[0][0].size // frame) * frame
        far = np.stack([p[0][:n] for p in pairs])
        near = np.stack([p[1][:n] for p in pairs])
        clean = synth_clean(near) if _fixture_has_clean(g) else None
        b = aecm.AecmBatch(S, fs, int(g["cng"]), int(g["echo_mode"]))
        rc, out = b.process_recordings_host(far, near, frame, ms, clean)
        assert [rc] == g["codes"].tolist(), f.name
        assert np.array_equal(out[0], g["out"]), f.name                 # stream 0 is the golden recording
        for k in (1, S - 1):
            s = aecm.Aecm()
            assert s.init(fs) == 0 and s.set_config(int(g["cng"]), int(g["echo_mode"])) == 0
            exp, _ = _run_session(s, far[k], near[k], frame, ms, None if clean is None else clean[k])
            assert np.array_equal(out[k], exp), (f.name, k)
            s.close()


def test_streaming_session_batch_ticks_equal_individual_sessions():
    """WebRtcAecmSessions_Tick: S sessions on a common 10 ms clock, including a jittering msInSndCardBuf
    and (last two cases) a nearendClean input."""
    rs = np.random.RandomState(9)
    for fs, frame, cng, em, with_clean in ((16000, 160, 1, 1, 0), (8000, 80, 1, 3, 0), (16000, 80, 0, 2, 0), (8000, 160, 1, 4, 0),
                                           (16000, 160, 1, 3, 1), (8000, 80, 1, 1, 1)):
        S, secs = 4, 5
        pairs = [synth_pair(70 + k, secs * fs // 64, fs, "mixed") for k in range(S)]
        n_ticks = pairs[0][0].size // frame
        far = np.stack([p[0][:n_ticks * frame] for p in pairs])
        near = np.stack([p[1][:n_ticks * frame] for p in pairs])
        clean = synth_clean(near) if with_clean else None
        ms_seq = [int(40 + rs.randint(-12, 13)) if i % 40 else int(rs.choice([-3, 0, 600, 90])) for i in range(n_ticks)]
        sb = aecm.AecmSessions(S, fs, cng, em)
        singles = []
        for k in range(S):
            s = aecm.Aecm()
            assert s.init(fs) == 0 and s.set_config(cng, em) == 0
            singles.append(s)
        for i in range(n_ticks):
            sl = slice(i * frame, (i + 1) * frame)
            rc, out = sb.tick_host(far[:, sl], near[:, sl], ms_seq[i], None if clean is None else clean[:, sl])
            assert rc in (0, aecm.ffi.AECM_BAD_PARAMETER_WARNING), (fs, i, rc)
            for k in (0, S - 1):
                assert singles[k].buffer_farend(far[k, sl]) == 0
                rc1, o1 = singles[k].process(near[k, sl], None if clean is None else clean[k, sl], ms_seq[i])
                assert rc == rc1 and np.array_equal(out[k], o1), (fs, frame, i, k)
        for s in singles:
            s.close()
        sb.close()


def test_streaming_ticks_device_pointers_unaligned_rows():
    """WebRtcAecmSessions_Tick on device pointers whose rows are NOT 8-byte aligned (odd stride, odd offset):
    the tick kernels must fall back from their 4-sample accesses; results equal the host-pointer path."""
    import torch
    fs, frame, S, n_ticks = 16000, 160, 6, 60
    pairs = [synth_pair(120 + k, n_ticks * frame // 64 + 1, fs, "mixed") for k in range(S)]
    far = np.stack([p[0][:n_ticks * frame] for p in pairs])
    near = np.stack([p[1][:n_ticks * frame] for p in pairs])
    ref = aecm.AecmSessions(S, fs, 1, 1)
    dev = aecm.AecmSessions(S, fs, 1, 1)
    stride = frame + 3                                             # odd row stride
    buf = torch.zeros((3, S * stride + 8), dtype=torch.int16, device="cuda")
    for i in range(n_ticks):
        sl = slice(i * frame, (i + 1) * frame)
        rc0, want = ref.tick_host(far[:, sl], near[:, sl], 40)
        off = 1 + (i % 2)                                          # odd / even element offsets
        for k, src in enumerate((far, near)):
            rows = torch.from_numpy(np.ascontiguousarray(src[:, sl])).cuda()
            buf[k, off:off + S * stride].view(S, stride)[:, :frame] = rows
        base = buf.data_ptr() + 2 * off
        row_bytes = buf.stride(0) * 2
        rc1 = dev.tick_device(base, base + row_bytes, base + 2 * row_bytes, stride, frame, 40)
        got = buf[2, off:off + S * stride].view(S, stride)[:, :frame].cpu().numpy()
        assert rc0 == rc1 and np.array_equal(got, want), i
    ref.close()
    dev.close()


def test_streaming_ticks_with_per_session_sound_card_delay():
    """WebRtcAecmSessions_TickPerSession: every session has its own msInSndCardBuf history (constant per
    session, out-of-range values, a mid-run change, and a uniform Tick mixed in); each must equal an
    individual WebRtcAecm_* session driven with the same values."""
    rs = np.random.RandomState(21)
    for fs, frame, with_clean in ((16000, 160, 0), (8000, 80, 1)):
        S, secs = 12, 4
        pairs = [synth_pair(90 + k, secs * fs // 64, fs, "mixed") for k in range(S)]
        n_ticks = pairs[0][0].size // frame
        far = np.stack([p[0][:n_ticks * frame] for p in pairs])
        near = np.stack([p[1][:n_ticks * frame] for p in pairs])
        clean = synth_clean(near) if with_clean else None
        base_ms = np.array([40, 40, 100, 250, -5, 700, 0, 40, 130, 100, 60, 40], dtype=np.int16)
        sb = aecm.AecmSessions(S, fs, 1, 3)
        singles = []
        for k in range(S):
            s = aecm.Aecm()
            assert s.init(fs) == 0 and s.set_config(1, 3) == 0
            singles.append(s)
        for i in range(n_ticks):
            sl = slice(i * frame, (i + 1) * frame)
            ms = base_ms.copy()
            if i >= n_ticks // 2:
                ms[1] = 90                                    # session 1 changes its delay report mid-run
            if i % 50 == 49:
                ms[7] = int(rs.randint(0, 400))               # session 7 jitters now and then
            c = None if clean is None else clean[:, sl]
            if i % 97 == 96:                                  # a uniform tick in between
                ms[:] = 55
                rc, out = sb.tick_host(far[:, sl], near[:, sl], 55, c)
                codes = np.full(S, rc)
            else:
                rc, out, codes = sb.tick_host_per_session(far[:, sl], near[:, sl], ms, c)
            assert rc in (0, aecm.ffi.AECM_BAD_PARAMETER_WARNING), (fs, i, rc)
            for k in range(S):
                assert singles[k].buffer_farend(far[k, sl]) == 0
                rc1, o1 = singles[k].process(near[k, sl], None if clean is None else clean[k, sl], int(ms[k]))
                if i % 97 != 96:
                    assert codes[k] == rc1, (fs, i, k)
                assert np.array_equal(out[k], o1), (fs, frame, i, k)
        for s in singles:
            s.close()
        sb.close()


def test_streaming_ticks_every_session_its_own_delay_in_one_tick():
    """Every session of a tick with its own msInSndCardBuf (the session wrapper runs per session on the device, so
    there is nothing to run out of), some of them out of range."""
    S, frame, fs = 1030, 160, 16000
    far, near = synth_pair(33, 40, fs, "mixed")
    far = np.tile(far[:10 * frame], (S, 1))
    near = np.tile(near[:10 * frame], (S, 1))
    sb = aecm.AecmSessions(S, fs, 1, 1)
    ones = {k: aecm.Aecm() for k in (0, 7, S - 1)}
    for one in ones.values():
        assert one.init(fs) == 0 and one.set_config(1, 1) == 0
    for i in range(10):
        sl = slice(i * frame, (i + 1) * frame)
        if i == 4:
            ms = np.arange(S, dtype=np.int16)                                                               # S distinct values
            rc, out, codes = sb.tick_host_per_session(far[:, sl], near[:, sl], ms)
            assert rc == aecm.ffi.AECM_BAD_PARAMETER_WARNING                                                # the sessions beyond 500 ms
            for k, one in ones.items():
                assert one.buffer_farend(far[k, sl]) == 0
                rc1, o1 = one.process(near[k, sl], None, int(ms[k]))
                assert codes[k] == rc1 and np.array_equal(out[k], o1), k
            continue
        rc, out = sb.tick_host(far[:, sl], near[:, sl], 40)
        for k, one in ones.items():
            assert one.buffer_farend(far[k, sl]) == 0
            rc1, o1 = one.process(near[k, sl], None, 40)
            assert rc == rc1 == 0 and np.array_equal(out[k], o1), (i, k)
    for one in ones.values():
        one.close()
    sb.close()


def _write_wav(path, rate, samples):
    import wave
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes(np.asarray(samples, dtype="<i2").tobytes())


def _read_wav(path):
    import wave
    with wave.open(str(path), "rb") as w:
        assert w.getnchannels() == 1 and w.getsampwidth() == 2
        return w.getframerate(), np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")


def test_cli_single_pair_and_batch_mode(tmp_path):
    """aecm_run: the reference CLI's procedure (main.cc) on one WAV pair and on a list of pairs."""
    import subprocess
    from webrtc_aecm_amd import build
    build.build()
    g = np.load(GOLDEN / "session_s7_fs16000_f160_c1_e1_ms40.npz")        # main.cc's parameters
    far, near = synth_pair(int(g["seed"]), int(g["n_blocks"]), 16000, "mixed")
    far = np.concatenate([far, np.zeros(57, np.int16)])                   # ragged tail: must stay untouched
    near = np.concatenate([near, np.arange(57, dtype=np.int16)])
    _write_wav(tmp_path / "far.wav", 16000, far)
    _write_wav(tmp_path / "near.wav", 16000, near)
    r = subprocess.run([str(build.CLI), str(tmp_path / "far.wav"), str(tmp_path / "near.wav")], capture_output=True, text=True)
    assert r.returncode == 0 and "time interval" in r.stdout, r.stdout + r.stderr
    rate, out = _read_wav(tmp_path / "near_out.wav")
    n = g["out"].size
    assert rate == 16000 and out.size == near.size
    assert np.array_equal(out[:n], g["out"]) and np.array_equal(out[n:], near[n:])
    # batch mode: three pairs, two rates, different lengths
    g8 = np.load(GOLDEN / "session_s8_fs8000_f80_c1_e3_ms40.npz")
    far8, near8 = synth_pair(int(g8["seed"]), int(g8["n_blocks"]), 8000, "mixed")
    _write_wav(tmp_path / "f8.wav", 8000, far8)
    _write_wav(tmp_path / "n8.wav", 8000, near8)
    _write_wav(tmp_path / "fs.wav", 16000, far[:48000])
    _write_wav(tmp_path / "ns.wav", 16000, near[:48000])
    (tmp_path / "pairs.txt").write_text(f"{tmp_path}/far.wav {tmp_path}/near.wav\n{tmp_path}/f8.wav {tmp_path}/n8.wav\n"
                                        f"{tmp_path}/fs.wav {tmp_path}/ns.wav\n")
    (tmp_path / "near_out.wav").unlink()
    r = subprocess.run([str(build.CLI), "--batch", str(tmp_path / "pairs.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    _, out = _read_wav(tmp_path / "near_out.wav")
    assert np.array_equal(out[:n], g["out"]) and np.array_equal(out[n:], near[n:])
    _, outs = _read_wav(tmp_path / "ns_out.wav")
    assert np.array_equal(outs, g["out"][:48000])                         # a prefix of a recording is a valid recording
    # the same list sharded over two device shards (two host threads, two engines; both on device 0 of this box)
    (tmp_path / "near_out.wav").unlink()
    (tmp_path / "ns_out.wav").unlink()
    r = subprocess.run([str(build.CLI), "--batch", str(tmp_path / "pairs.txt"), "--devices", "0,0"], capture_output=True, text=True)
    assert r.returncode == 0 and "2 device shard(s)" in r.stdout, r.stdout + r.stderr
    _, out2 = _read_wav(tmp_path / "near_out.wav")
    _, outs2 = _read_wav(tmp_path / "ns_out.wav")
    assert np.array_equal(out2, out) and np.array_equal(outs2, outs)
    # the 8 kHz pair ran with the CLI's echoMode 1, not the fixture's 3: check against a live session instead
    _, out8 = _read_wav(tmp_path / "n8_out.wav")
    s = aecm.Aecm()
    assert s.init(8000) == 0 and s.set_config(1, 1) == 0
    n8 = (near8.size // 80) * 80
    exp8, _ = _run_session(s, far8[:n8], near8[:n8], 80, 40)
    assert np.array_equal(out8[:n8], exp8)


def test_session_abi_error_codes():
    s = aecm.Aecm()
    z = np.zeros(160, dtype=np.int16)
    assert s.buffer_farend(z) == aecm.ffi.AECM_UNINITIALIZED_ERROR
    assert s.process(z)[0] == aecm.ffi.AECM_UNINITIALIZED_ERROR
    assert s.set_config(1, 3) == aecm.ffi.AECM_UNINITIALIZED_ERROR
    assert s.init(44100) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    assert s.init(16000) == 0
    assert s.buffer_farend(z[:100]) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    assert s.process(z[:100])[0] == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    assert s.set_config(2, 3) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    assert s.set_config(1, 5) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    assert s.process(z, None, -5)[0] == aecm.ffi.AECM_BAD_PARAMETER_WARNING
    assert s.init_echo_path(z[:64]) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    lib = aecm.load()
    assert lib.WebRtcAecm_Init(None, 16000) == -1
    assert lib.WebRtcAecm_echo_path_size_bytes() == 130
    lib.WebRtcAecm_Free(None)


@pytest.mark.skipif(not pyoracle.have_reference(), reason="prebuilt oracle/_ref/libaecm_ref.so not present")
def test_block_parity_vs_real_reference():
    S, T, fs = 8, 1500, 16000
    seeds = list(range(2000, 2000 + S))
    far, near = synth_streams(seeds, T, fs)
    b = aecm.AecmBatch(S, fs, 1, 1)
    out = b.process_host(far, near)
    for s in range(S):
        r = pyoracle.RefCoreStream(fs, 1, 1)
        assert np.array_equal(out[s], r.process(far[s], near[s])), f"stream {s}"
        assert np.array_equal(b.digest(s), r.digest())


def test_config2_4096_streams_bit_exact():
    """BASELINE config 2: 4096 streams, 16 kHz, every stream checked against the CPU checker -- the unmodified reference
    (WebRtcAecm_ProcessBlock, aecm_core_c.cc:368-711) where oracle/_ref is present, our restatement otherwise."""
    S, T, fs = 4096, 2048, 16000                 # SURVEY 8.d Config 2: T >= 2 048
    seeds = list(range(10000, 10000 + S))
    cfgs = [(1, 3)] * S
    far, near = synth_streams(seeds, T, fs)
    b = aecm.AecmBatch(S, fs)
    out = b.process_host(far, near)
    exp_out, exp_dig = oracle_batch(seeds, T, fs, cfgs, pairs=(far, near), prefer_reference=True)     # the reference itself when oracle/_ref travelled
    bad = [s for s in range(S) if not np.array_equal(out[s], exp_out[s])]
    assert not bad, f"{len(bad)} streams differ, first {bad[:5]}"
    for s in range(0, S, 97):
        assert np.array_equal(b.digest(s), exp_dig[s])


def _replicated_full_size_run(S, T, fs, U, seed0, chunking=None):
    """S streams replicating U distinct seeded (far, near) pairs, one launch of T blocks with everything resident in HBM
    (replication, the launch and the comparison all on the device: the buffers are several GB each).  Every stream's
    output must equal the CPU checker's answer for the pair it replicates (the unmodified reference where oracle/_ref is
    present, else the restatement), and sampled streams' state digests too."""
    import torch
    seeds = list(range(seed0, seed0 + U))
    far, near = synth_streams(seeds, T, fs)
    exp_out, exp_dig = oracle_batch(seeds, T, fs, [(1, 3)] * U, pairs=(far, near), prefer_reference=True)    # the reference itself when it travelled
    assert S % U == 0
    L = T * 64
    dfar = torch.from_numpy(far).cuda().unsqueeze(0).expand(S // U, U, L).reshape(S, L)       # stream s replicates pair s % U
    dnear = torch.from_numpy(near).cuda().unsqueeze(0).expand(S // U, U, L).reshape(S, L)
    dout = torch.empty_like(dnear)
    assert dfar.is_contiguous() and dout.numel() == S * L
    torch.cuda.synchronize()
    b = aecm.AecmBatch(S, fs)
    if chunking is not None:
        b.set_launch_chunking(*chunking)
    b.process_device(dfar.data_ptr(), dnear.data_ptr(), dout.data_ptr(), L, 64, T)
    b.synchronize()
    dexp = torch.from_numpy(exp_out).cuda()
    bad = 0
    for g0 in range(0, S // U, 64):                                   # compare in slabs (a 4.6 GB bool tensor otherwise)
        g1 = min(S // U, g0 + 64)
        bad += int((dout.view(S // U, U, L)[g0:g1] != dexp.unsqueeze(0)).sum().item())
    assert bad == 0, f"{bad} samples differ"
    for s in (0, U - 1, U, S // 2 + 5, S - 1):
        assert np.array_equal(b.digest(s), exp_dig[s % U]), f"digest of stream {s}"
    b.close()
    return S * L


def test_large_batch_properties_65536_streams():
    """BASELINE config 3 at full size AND past start-up: 65 536 streams x 1 100 blocks -- the start-up state changes at
    512 and 1 024 blocks (reference aecm_core_c.cc:420-424) and 65 536 x 1 100 x 64 = 4.6 G int16 elements per buffer, so
    element offsets above 2^32 are addressed (the bench workload's regime).  Checked through a size-independent property:
    replicated streams must equal the oracle's answer for the 64 distinct seeds they replicate."""
    n = _replicated_full_size_run(65536, 1100, 16000, 64, 400)
    assert n > 2 ** 32


def test_large_batch_one_wave_per_stream_form():
    """Launches larger than the chip take the chunk-queue kernel by default (the test above); the one-stream-per-wave form
    of the same launch (WebRtcAecmBatch_SetLaunchChunking(b, 0, ..)) at a size past start-up and past 2^32 elements."""
    n = _replicated_full_size_run(65536, 1056, 16000, 64, 400, chunking=(0, -1))
    assert n > 2 ** 32


@pytest.mark.parametrize("fs,clean", [(16000, False), (8000, False), (16000, True)])
def test_chunk_queue_launch_under_contention(fs, clean):
    """The chunk-queue kernel (aecm_block_kernels.hip) forced onto a batch far smaller than the chip: 24 streams cut into
    chunks of 4, 32 and 64 blocks, so that the wave that claims a stream's next chunk usually finds its predecessor still
    running and waits for it -- the hand-over of a stream's state between waves (lane vectors, scalars and far-spectrum
    history rows through memory at agent scope) is on the critical path of every item.  Outputs and complete state must
    equal the oracle's; two launches in a row continue each other."""
    S, T = 24, 192
    seeds = list(range(7100, 7100 + S))
    cfgs = [stream_config(s) for s in range(S)]
    far, near = synth_streams(seeds, 2 * T, fs)
    cln = synth_clean(near) if clean else None
    exp = []
    for s in range(S):
        o = pyoracle.OracleStream(fs, *cfgs[s])
        if clean:
            e = np.concatenate([o.process_block_clean(far[s, k * 64:(k + 1) * 64], near[s, k * 64:(k + 1) * 64], cln[s, k * 64:(k + 1) * 64])
                                for k in range(2 * T)])
        else:
            e = o.process(far[s], near[s])
        exp.append((e, o.digest()))
    for chunk in (4, 32, 64):
        b = aecm.AecmBatch(S, fs)
        for s, (cng, em) in enumerate(cfgs):
            b.set_config(cng, em, s, 1)
        b.set_launch_chunking(chunk, 0)
        assert b.describe_launch(T, clean) == (2, chunk)                # an explicit chunk length is taken as it is
        out = np.concatenate([b.process_host(far[:, :T * 64], near[:, :T * 64], cln[:, :T * 64] if clean else None),
                              b.process_host(far[:, T * 64:], near[:, T * 64:], cln[:, T * 64:] if clean else None)], axis=1)
        for s in range(S):
            assert np.array_equal(out[s], exp[s][0]), (chunk, s)
            assert np.array_equal(b.digest(s), exp[s][1]), (chunk, s, describe_digest_diff(b.digest(s), exp[s][1]))
        b.close()


@pytest.mark.parametrize("fs", [16000, 8000])
def test_pipelined_launch_sizes_and_lengths(fs):
    """The pipelined form of launches the chip holds at once (aecm_block_kernels.hip: aecm_process_pipelined_kernel): batch
    sizes that leave workgroups partly empty (1..9, 333 streams: a back wave or a front wave's second stream without a
    stream), launches of 1, 2 and 3 blocks (the front waves' prologue and the last, empty trip), and launches that
    continue each other across forms (pipelined, one wavefront per stream, pipelined): outputs and complete state."""
    T_parts = (1, 2, 3, 130, 64)
    T = sum(T_parts)
    cus = aecm.device_info(0)[1]
    # (round 6) sizes between the shapes' full loads: every CU then gets its full count of workgroups, of one to four streams
    # each -- 2 x cus + 88: sixteen-wave workgroups of 3 and 2; 6 x cus: ten-wave workgroups of 3 (1 536 streams on an MI355X);
    # 10 x cus: eight-wave workgroups of 4 and 3 (2 560); 14 x cus - 3: six-wave workgroups of 4 and 3
    sizes = (1, 2, 3, 4, 5, 7, 9, 333, 2 * cus + 88, 6 * cus, 10 * cus, 14 * cus - 3) if fs == 16000 else (1, 3, 5, 333, 5 * cus + 1)
    for S in sizes:
        seeds = list(range(7500, 7500 + min(S, 24)))
        far, near = synth_streams(seeds, T, fs)
        exp = []
        for k in range(len(seeds)):
            o = pyoracle.OracleStream(fs, *stream_config(k))
            exp.append((o.process(far[k], near[k]), o.digest()))
        idx = np.arange(S) % len(seeds)
        far_s, near_s = far[idx], near[idx]
        b = aecm.AecmBatch(S, fs)
        for s in range(S):
            b.set_config(*stream_config(int(idx[s])), s, 1)
        outs, pos = [], 0
        for i, t in enumerate(T_parts):
            b.set_launch_pipelining(0 if i == 3 else 1)
            assert b.describe_launch(t)[0] == (0 if i == 3 else 3)
            outs.append(b.process_host(far_s[:, pos * 64:(pos + t) * 64], near_s[:, pos * 64:(pos + t) * 64]))
            pos += t
        out = np.concatenate(outs, axis=1)
        want = np.stack([e[0] for e in exp])[idx]
        bad = np.nonzero((out != want).any(axis=1))[0]
        assert bad.size == 0, (S, bad[:8].tolist())
        for s in (range(S) if S < 400 else list(range(0, S, S // 97)) + [S - 1]):
            assert np.array_equal(b.digest(s), exp[idx[s]][1]), (S, s, describe_digest_diff(b.digest(s), exp[idx[s]][1]))
        b.close()


@pytest.mark.parametrize("wish,shape", [({}, 0x1a02),                                            # by size: sixteen waves (delay + gain waves)
                                        ({"pipe_gain_waves": 0}, 0x802),                        # twelve: delay waves, one per stream
                                        ({"pipe_delay_waves": 0}, 0x2),                         # eight: two tail waves
                                        ({"pipe_delay_waves": 0, "pipe_raw": 1}, 0x402),
                                        ({"pipe_delay_waves": 0, "pipe_raw": 1, "pipe_front_waves": 4}, 0x602),
                                        ({"pipe_tail_waves": 0}, 0x0),                          # six: the two-role form
                                        # workgroups of four streams, the last one partly empty (the form of rounds 4 and 5)
                                        ({"pipe_spread": 0}, 0x1a02),
                                        ({"pipe_spread": 0, "pipe_delay_waves": 0, "pipe_raw": 1, "pipe_front_waves": 4}, 0x602),
                                        ({"pipe_spread": 0, "pipe_tail_waves": 0}, 0x0),
                                        # every slot rotation at its other values (any bijection of waves to slots is correct)
                                        ({"pipe_rot": 2 | (3 << 2) | (1 << 4) | (1 << 6) | (1 << 8)}, 0x1a02),
                                        ({"pipe_spread": 0, "pipe_rot": 3 | (2 << 2) | (2 << 4) | (3 << 6) | (3 << 8)}, 0x1a02),
                                        ({"pipe_spread": 0, "pipe_gain_waves": 0, "pipe_rot": 1 | (3 << 4) | (1 << 8)}, 0x802),
                                        ({"pipe_spread": 0, "pipe_delay_waves": 0, "pipe_raw": 1, "pipe_rot": 1 | (2 << 6) | (3 << 8)}, 0x402),
                                        ({"pipe_spread": 0, "pipe_tail_waves": 0, "pipe_rot": 3 | (1 << 6)}, 0x0)])
def test_pipelined_shapes_by_wish(wish, shape):
    """Every instantiation of the pipelined kernel a small launch can be given (wishes on the batch's launch policy:
    WebRtcAecmBatch_SetLaunchPolicy): 11 streams (one per workgroup: the other slots' waves only keep the barriers; with
    pipe_spread = 0 two workgroups of four and one of three, a two-stream wave with one of its two) through launches of 1, 2,
    3, 4, 5 and 150 blocks -- each role's idle steps in front of and behind its block loop, the slot rings of two, three and four
    steps, the fixed-delay configuration (the delay waves' AlignedFarend then never uses the estimate) -- outputs and complete
    state against the oracle."""
    fs, S, T_parts = 16000, 11, (1, 2, 3, 4, 5, 150)
    T = sum(T_parts)
    seeds = list(range(7700, 7700 + S))
    far, near = synth_streams(seeds, T, fs)
    b = aecm.AecmBatch(S, fs)
    b.set_launch_pipelining(2)                               # (a threshold set through the ABI: launches of any length, not only of three blocks and more)
    b.set_launch_policy(**wish)
    exp = []
    for k in range(S):
        o = pyoracle.OracleStream(fs, *stream_config(k))
        b.set_config(*stream_config(k), k, 1)
        if k in (2, 5, 6):                                   # a fixed delay of 0 / 1 / 7 blocks, the last with the NLP off
            fixed, nlp = {2: (0, 1), 5: (1, 1), 6: (7, 0)}[k]
            o.control(fixed, nlp)
            b.control(fixed, nlp, k, 1)
        exp.append((o.process(far[k], near[k]), o.digest()))
    outs, pos = [], 0
    for t in T_parts:
        assert b.describe_launch(t) == (3, shape)
        outs.append(b.process_host(far[:, pos * 64:(pos + t) * 64], near[:, pos * 64:(pos + t) * 64]))
        pos += t
    out = np.concatenate(outs, axis=1)
    for k in range(S):
        assert np.array_equal(out[k], exp[k][0]), (wish, k)
        assert np.array_equal(b.digest(k), exp[k][1]), (wish, k, describe_digest_diff(b.digest(k), exp[k][1]))
    b.close()


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_random_launch_policies_never_change_results(seed):
    """Results never depend on how a launch is scheduled: one batch advanced by a dozen launches of random lengths, each under a
    random valid launch policy -- any pipelined shape, workgroups of four or spread, any slot rotation and role priority, the chunk
    queue forced onto small batches, one wavefront per stream -- must produce the oracle's outputs and states for every stream."""
    rs = np.random.RandomState(seed)
    cus = aecm.device_info(0)[1]
    fs = int(rs.choice([16000, 8000]))
    S = int(rs.choice([rs.randint(2, 40), rs.randint(40, 700), cus + rs.randint(-3, 4), 2 * cus + rs.randint(1, 90)]))
    U = min(S, 20)
    parts = [int(rs.choice([1, 2, 3, 5, 17, 40, 64])) for _ in range(12)]
    T = sum(parts)
    seeds = list(range(8800 + seed * 50, 8800 + seed * 50 + U))
    far_u, near_u = synth_streams(seeds, T, fs)
    exp = []
    for k in range(U):
        o = pyoracle.OracleStream(fs, *stream_config(k))
        exp.append((o.process(far_u[k], near_u[k]), o.digest()))
    idx = np.arange(S) % U
    far, near = far_u[idx], near_u[idx]
    b = aecm.AecmBatch(S, fs)
    for s_ in range(S):
        b.set_config(*stream_config(int(idx[s_])), s_, 1)
    outs, pos, forms = [], 0, set()
    for t in parts:
        wishes = dict(pipe_tail_waves=int(rs.choice([-1, 0, 2])), pipe_front_waves=int(rs.choice([-1, 2, 4])), pipe_raw=int(rs.choice([-1, 0, 1])),
                      pipe_delay_waves=int(rs.choice([-1, 0, 2, 4])), pipe_gain_waves=int(rs.choice([-1, 0, 4])), pipe_spread=int(rs.randint(0, 2)),
                      pipe_rot=int(rs.choice([-1, rs.randint(0, 1024)])), pipe_prio=int(rs.choice([-1, rs.randint(0, 256)])),
                      pipelined_min_streams=int(rs.choice([2, 2, 2, 0])), pipelined_min_blocks=int(rs.choice([1, 3])))
        if rs.rand() < 0.25:                                  # the chunk queue on a batch this small: chunks of 8, at least two of them
            wishes.update(queue_min_streams=0, queue_chunk_blocks=8, queue_chunk_explicit=1, pipelined_min_streams=0)
        else:
            wishes.update(queue_min_streams=-1, queue_chunk_blocks=128, queue_chunk_explicit=0)
        b.set_launch_policy(**wishes)
        forms.add(b.describe_launch(t)[0])
        outs.append(b.process_host(far[:, pos * 64:(pos + t) * 64], near[:, pos * 64:(pos + t) * 64]))
        pos += t
    out = np.concatenate(outs, axis=1)
    want = np.stack([e[0] for e in exp])[idx]
    bad = np.nonzero((out != want).any(axis=1))[0]
    assert bad.size == 0, (seed, S, fs, parts, bad[:8].tolist())
    for s_ in (range(S) if S < 100 else list(range(0, S, S // 37)) + [S - 1]):
        assert np.array_equal(b.digest(s_), exp[idx[s_]][1]), (seed, S, s_, describe_digest_diff(b.digest(s_), exp[idx[s_]][1]))
    assert 3 in forms
    b.close()


def test_chunk_queue_launch_the_chip_holds_at_once():
    """5 003 streams (between the pipelined form's 4 096 and the chip's 7 168 resident waves): every wave is resident from the
    start and the launch still takes the chunk queue (items of 32 blocks), because the waves dispatched first pull ahead and
    then take more of the work.  Every stream against the oracle's answer for the pair it replicates, with a clean input too."""
    import torch
    S, T, fs, U = 5003, 200, 16000, 16
    seeds = list(range(7900, 7900 + U))
    far, near = synth_streams(seeds, T, fs)
    idx = torch.arange(S) % U
    for clean in (False, True):
        cln = synth_clean(near) if clean else None
        exp = []
        for k in range(U):
            o = pyoracle.OracleStream(fs, 1, 3)
            exp.append(np.concatenate([o.process_block_clean(far[k, j * 64:(j + 1) * 64], near[k, j * 64:(j + 1) * 64], cln[k, j * 64:(j + 1) * 64])
                                       for j in range(T)]) if clean else o.process(far[k], near[k]))
        dfar = torch.from_numpy(far).cuda()[idx].contiguous()
        dnear = torch.from_numpy(near).cuda()[idx].contiguous()
        dclean = torch.from_numpy(cln).cuda()[idx].contiguous() if clean else None
        dout = torch.empty_like(dnear)
        b = aecm.AecmBatch(S, fs)
        assert b.describe_launch(T, clean) == (2, 32)
        b.process_device(dfar.data_ptr(), dnear.data_ptr(), dout.data_ptr(), T * 64, 64, T, dclean.data_ptr() if clean else None)
        b.synchronize()
        assert int((dout != torch.from_numpy(np.stack(exp)).cuda()[idx]).sum().item()) == 0, clean
        b.close()


def test_chunk_queue_half_a_million_hand_overs():
    """The state hand-over between waves (memory at agent scope, any CU of any XCD picks up a stream's next chunk) as often
    as a test can afford: 8 200 streams x 512 blocks in chunks of 8 -- 64 hand-overs per stream, 525 000 in the launch, the
    chip full throughout -- against the same batch run with one wavefront per stream (itself pinned to the oracle and the
    reference elsewhere): every output sample and every stream's complete state digest."""
    import torch
    S, T, fs, U = 8200, 512, 16000, 40
    seeds = list(range(7700, 7700 + U))
    far, near = synth_streams(seeds, T, fs)
    idx = torch.arange(S) % U
    dfar = torch.from_numpy(far).cuda()[idx].contiguous()
    dnear = torch.from_numpy(near).cuda()[idx].contiguous()
    outs, digs = [], []
    for chunking in ((8, -1), (0, -1)):
        b = aecm.AecmBatch(S, fs, echo_mode=2)
        b.set_launch_chunking(*chunking)
        assert b.describe_launch(T)[0] == (2 if chunking[0] else 1)
        dout = torch.empty_like(dnear)
        b.process_device(dfar.data_ptr(), dnear.data_ptr(), dout.data_ptr(), T * 64, 64, T)
        b.synchronize()
        outs.append(dout)
        digs.append(np.stack([b.digest(s) for s in range(0, S, 7)]))
        b.close()
    assert int((outs[0] != outs[1]).sum().item()) == 0
    assert np.array_equal(digs[0], digs[1])
    # and the one-wave run is the oracle's
    o = pyoracle.OracleStream(fs, 1, 2)
    assert np.array_equal(outs[1][5].cpu().numpy(), o.process(far[5], near[5]))


def test_launch_form_by_size():
    """Which kernel a launch takes (WebRtcAecmBatch_DescribeLaunch; INTEGRATION.md has the table): one stream -> one wavefront
    per stream; 2 .. 4 x 4 x CUs streams -> pipelined (not with a clean input, not the safe variant; with two tail waves per
    workgroup up to 3 x 4 x CUs streams -- and delay and gain waves up to 4 x CUs --, launches of three blocks and more, above that without them and -- in launches long enough --
    balanced); more -> chunk queue if
    the launch is at least two chunks long (chunks of 32 blocks up to the chip's resident waves, of 128 above), else one
    wavefront per stream."""
    cus = aecm.device_info(0)[1]
    pipe_max, resident, rotation = cus * 16, cus * 28, cus * 24
    tail_max = cus * 12                       # eight-wave workgroups (two tail waves) come three to a CU
    # pipelined launches: the second value is the shape (tail waves | 0x100 balanced | 0x200 four front waves | 0x400 raw hand-over | 0x800 delay
    # waves | 0x1000 gain waves)
    for S, T, clean, want, want_chunk in ((1, 300, False, 0, 0), (2, 300, False, 3, 0x1a02), (cus * 4, 300, False, 3, 0x1a02), (cus * 4 + 1, 300, False, 3, 0x1a02),
                                          (cus * 8, 300, False, 3, 0x1a02), (cus * 8 + 1, 300, False, 3, 0x402),
                                          (tail_max, 300, False, 3, 0x402), (tail_max + 1, 300, False, 3, 0x500),
                                          (pipe_max, 3, False, 3, 0), (pipe_max, 2, False, 0, 0), (cus * 4, 2, False, 0, 0), (cus * 4, 3, False, 3, 0x1a02),
                                          (pipe_max, 300, False, 3, 0x500), (pipe_max, 300, True, 0, 0),
                                          (pipe_max + 1, 300, False, 2, 32), (pipe_max + 1, 63, False, 0, 0), (rotation + 1, 63, False, 1, 0),
                                          (resident, 64, True, 2, 32), (resident + 1, 255, False, 1, 0), (resident + 1, 256, False, 2, 128),
                                          (resident + 1, 256, True, 2, 128)):
        b = aecm.AecmBatch(S, 16000)
        form, chunk = b.describe_launch(T, clean)
        assert (form, chunk) == (want, want_chunk), (S, T, clean, form, chunk)
        assert b.launch_policy().as_dict() == aecm.default_launch_policy(cus).as_dict()       # derived from the device's CU count alone
        if S == pipe_max:
            b.set_variant(aecm.KERNEL_SAFE)
            assert b.describe_launch(T, clean)[0] == 0
            b.set_variant(aecm.KERNEL_FAST)
            b.set_launch_pipelining(0)
            assert b.describe_launch(T, clean)[0] == 0
        if S == resident + 1 and T == 256:
            b.set_launch_chunking(0)
            assert b.describe_launch(T, clean) == (1, 0)
            b.set_launch_chunking(64, 10)
            assert b.describe_launch(T, clean) == (2, 64)
        b.close()


def test_chunk_queue_launch_larger_than_the_chip():
    """9 001 streams (more than the chip holds waves, not a multiple of a workgroup's four) x 300 blocks in chunks of 128,
    128 and 44: the launch takes the queue form by itself.  Every stream must equal the oracle's answer for the pair it
    replicates; then the same batch continues with a launch in the one-stream-per-wave form and must still agree."""
    import torch
    S, T, fs, U = 9001, 300, 16000, 16
    seeds = list(range(7300, 7300 + U))
    far, near = synth_streams(seeds, 2 * T, fs)
    exp_out, exp_dig = oracle_batch(seeds, 2 * T, fs, [(1, 3)] * U, pairs=(far, near), prefer_reference=True)
    idx = torch.arange(S) % U
    dfar = torch.from_numpy(far).cuda()[idx].contiguous()
    dnear = torch.from_numpy(near).cuda()[idx].contiguous()
    dout = torch.empty_like(dnear)
    dexp = torch.from_numpy(exp_out).cuda()[idx]
    torch.cuda.synchronize()
    L = 2 * T * 64
    b = aecm.AecmBatch(S, fs)
    b.process_device(dfar.data_ptr(), dnear.data_ptr(), dout.data_ptr(), L, 64, T)
    b.set_launch_chunking(0, -1)
    half = T * 128                                                       # bytes into every row
    b.process_device(dfar.data_ptr() + half, dnear.data_ptr() + half, dout.data_ptr() + half, L, 64, T)
    b.synchronize()
    assert int((dout != dexp).sum().item()) == 0
    for s in (0, 15, 16, 4500, 9000):
        assert np.array_equal(b.digest(s), exp_dig[s % U]), s
    b.close()


def test_state_snapshot_import_is_validated():
    """ImportState refuses blobs that are not snapshots of this layout or whose index-like fields are out of
    range (they would be used as addresses by the kernel), and leaves the stream untouched."""
    import struct
    fs = 16000
    far, near = synth_streams([5], 200, fs)
    b = aecm.AecmBatch(1, fs)
    b.process_host(far, near)
    blob = b.export_state(0)
    lib = aecm.load()
    dig = b.digest(0)

    def imp(x):
        return lib.WebRtcAecmBatch_ImportState(b.h, 0, bytes(x), len(x))
    assert imp(blob) == 0
    bad = bytearray(blob)
    bad[0] ^= 0xff                                                        # magic
    assert imp(bad) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    bad = bytearray(blob)
    struct.pack_into("<I", bad, 4, 99)                                    # layout version
    assert imp(bad) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    scal0 = 32 + 12 * 64 * 4                                              # header + lane vectors
    for field, value in ((3, 1000), (3, -1), (28, 100), (29, 3), (2, 7)):  # S_HISTPOS, S_LAST_DELAY, S_MULT, S_STARTUP
        bad = bytearray(blob)
        struct.pack_into("<i", bad, scal0 + 4 * field, value)
        assert imp(bad) == aecm.ffi.AECM_BAD_PARAMETER_ERROR, (field, value)
    assert lib.WebRtcAecmBatch_ImportState(b.h, 0, blob[:-2], len(blob) - 2) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    assert np.array_equal(b.digest(0), dig)
    # a stream of the other rate may be imported, but whole-recording scheduling is then refused until Init
    b8 = aecm.AecmBatch(1, 8000)
    assert lib.WebRtcAecmBatch_ImportState(b8.h, 0, blob, len(blob)) == 0
    z = np.zeros((1, 1600), np.int16)
    rc, _ = b8.process_recordings_host(z, z, 160)
    assert rc == aecm.ffi.AECM_UNSUPPORTED_FUNCTION_ERROR
    assert lib.WebRtcAecmBatch_Control(b.h, 100, 1, 0, -1) == aecm.ffi.AECM_BAD_PARAMETER_ERROR       # delay beyond the 100-slot history


def test_session_jitter_goldens():
    """Committed reference outputs for a jittering msInSndCardBuf + far-end underruns (tools/gen_golden.py,
    sessjit_*): the single-session ABI and the streaming ticks (uniform + per-session forms) must reproduce them."""
    files = golden_files("sessjit_")
    assert len(files) >= 3
    for f in files:
        g = np.load(f)
        fs, frame, cng, em = int(g["fs"]), int(g["frame"]), int(g["cng"]), int(g["echo_mode"])
        far, near = synth_pair(int(g["seed"]), int(g["n_blocks"]), fs, "mixed")
        ms_seq, far_present = g["ms_seq"], g["far_present"]
        assert np.array_equal(ms_seq, call_pattern(int(g["seed"]), ms_seq.size)[0])
        s = aecm.Aecm()
        assert s.init(fs) == 0 and s.set_config(cng, em) == 0
        out, codes = drive_session(s, far, near, frame, ms_seq, far_present)
        s.close()
        assert np.array_equal(codes, g["codes"]) and np.array_equal(out, g["out"]), f.name
        # the same as session 1 of a 3-session streaming batch (sessions 0 / 2 run a plain pattern)
        S = 3
        sb = aecm.AecmSessions(S, fs, cng, em)
        n_calls = ms_seq.size
        got = np.empty_like(out)
        for i in range(n_calls):
            sl = slice(i * frame, (i + 1) * frame)
            ms = np.array([40, ms_seq[i], 40], dtype=np.int16)
            fl = np.array([0, 0 if far_present[i] else aecm.ffi.SESSION_NO_FAREND, 0], dtype=np.uint8)
            rc, o, c = sb.tick_host_per_session(np.stack([far[sl]] * S), np.stack([near[sl]] * S), ms, flags=fl)
            assert c[1] == g["codes"][i], (f.name, i)
            got[sl] = o[1]
        sb.close()
        assert np.array_equal(got, g["out"]), f.name


_needs_ref = pytest.mark.skipif(not pyoracle.have_reference(), reason="prebuilt oracle/_ref/libaecm_ref.so not present")


@_needs_ref
def test_single_session_abi_vs_reference_jitter_underruns_all_call_sizes():
    """WebRtcAecm_* on the GPU against the reference's own session ABI driven call by call with the same hostile
    pattern: jittering / out-of-range msInSndCardBuf, far-end underruns, 80- and 160-sample calls, both rates,
    with and without nearendClean."""
    for k, (fs, frame, with_clean) in enumerate(((16000, 160, 0), (16000, 80, 0), (8000, 80, 1), (8000, 160, 0), (16000, 160, 1))):
        far, near = synth_pair(500 + k, 4 * fs // 64, fs, "mixed")
        clean = synth_clean(near) if with_clean else None
        n_calls = far.size // frame
        ms_seq, far_present = call_pattern(40 + k, n_calls)
        r = pyoracle.RefSession(fs, 1, 2)
        s = aecm.Aecm()
        assert s.init(fs) == 0 and s.set_config(1, 2) == 0
        exp, exp_codes = drive_session(r, far, near, frame, ms_seq, far_present, clean)
        got, codes = drive_session(s, far, near, frame, ms_seq, far_present, clean)
        s.close()
        assert np.array_equal(codes, exp_codes), (fs, frame)
        assert np.array_equal(got, exp), (fs, frame, int(np.nonzero(got != exp)[0][0]) // frame)


@_needs_ref
def test_streaming_ticks_vs_reference_sessions():
    """WebRtcAecmSessions_Tick / TickPerSession / TickFlags against one reference session per stream (not against
    our own single-session path): uniform jittering delay, then per-session delays with underruns."""
    for fs, frame, with_clean in ((16000, 160, 0), (8000, 80, 1), (8000, 160, 0)):
        S = 6
        pairs = [synth_pair(640 + k, 4 * fs // 64, fs, "mixed") for k in range(S)]
        n_calls = pairs[0][0].size // frame
        far = np.stack([p[0][:n_calls * frame] for p in pairs])
        near = np.stack([p[1][:n_calls * frame] for p in pairs])
        clean = synth_clean(near) if with_clean else None
        pats = [call_pattern(70 + k, n_calls) for k in range(S)]
        refs = [pyoracle.RefSession(fs, 1, 3) for _ in range(S)]
        sb = aecm.AecmSessions(S, fs, 1, 3)
        for i in range(n_calls):
            sl = slice(i * frame, (i + 1) * frame)
            c = None if clean is None else clean[:, sl]
            if i < n_calls // 3:                                   # uniform tick, jittering value
                ms = np.full(S, pats[0][0][i], dtype=np.int16)
                fl = np.zeros(S, dtype=np.uint8)
                rc, out = sb.tick_host(far[:, sl], near[:, sl], int(ms[0]), c)
                codes = np.full(S, rc)
            else:                                                  # per-session values + underruns
                ms = np.array([pats[k][0][i] for k in range(S)], dtype=np.int16)
                fl = np.array([0 if pats[k][1][i] else aecm.ffi.SESSION_NO_FAREND for k in range(S)], dtype=np.uint8)
                fl[0] = 0
                rc, out, codes = sb.tick_host_per_session(far[:, sl], near[:, sl], ms, c, flags=fl)
            for k in range(S):
                if not fl[k]:
                    assert refs[k].buffer_farend(far[k, sl]) == 0
                rc1, o1 = refs[k].process(near[k, sl], None if clean is None else clean[k, sl], int(ms[k]))
                assert codes[k] == rc1, (fs, frame, i, k)
                assert np.array_equal(out[k], o1), (fs, frame, i, k)
        sb.close()


def test_session_burst_goldens():
    """sessburst_* fixtures (reference outputs for far-end bursts, jitter-buffer overflow and mid-session set_config /
    InitEchoPath / re-Init at the other rate) on WebRtcAecm_* on the GPU."""
    files = golden_files("sessburst_")
    assert len(files) >= 2
    for f in files:
        g = np.load(f)
        s = aecm.Aecm()
        out, codes, paths = run_burst_fixture(s, g)
        s.close()
        assert np.array_equal(codes, g["codes"]) and np.array_equal(out, g["out"]) and np.array_equal(paths, g["paths"]), f.name


@_needs_ref
@pytest.mark.parametrize("fs,frame", [(16000, 160), (8000, 80), (16000, 80), (8000, 160)])
def test_single_session_abi_far_end_bursts_and_mid_session_reconfiguration(fs, frame):
    """WebRtcAecm_* on the GPU against the reference's ABI call by call: k = 0, 1, 1, 1, 2, 3 WebRtcAecm_BufferFarend calls
    per WebRtcAecm_Process, a 30-frame burst every 50 calls (the 4 000-sample jitter buffer truncates: reference
    ring_buffer.c:142-170, echo_control_mobile.cc:215-234), calls without any far frame, and in between
    WebRtcAecm_set_config (valid and refused), InitEchoPath / GetEchoPath, WebRtcAecm_Init at the other rate and back."""
    n_calls = 3 * fs // frame if frame == 160 or fs == 8000 else 400
    ms_seq, far_calls = call_pattern(7 + fs // 8000 + frame, n_calls, bursts=True)
    far, _ = synth_pair(61, far_frames_needed(far_calls) * frame // 64 + 1, fs, "mixed")
    _, near = synth_pair(61, n_calls * frame // 64 + 1, fs, "mixed")
    events = reconfiguration_events(fs, n_calls)
    r = pyoracle.RefSession(fs, 1, 3)
    s = aecm.Aecm()
    assert s.init(fs) == 0 and s.set_config(1, 3) == 0
    exp, exp_codes = drive_session(r, far, near, frame, ms_seq, far_calls, events=events)
    got, codes = drive_session(s, far, near, frame, ms_seq, far_calls, events=events)
    s.close()
    assert np.array_equal(codes, exp_codes)
    assert np.array_equal(got, exp), int(np.nonzero(got != exp)[0][0]) // frame
    assert len(r.event_log) == 3 and all(np.array_equal(a, b) for a, b in zip(r.event_log, s.event_log))


@_needs_ref
@pytest.mark.parametrize("fs,frame,with_clean", [(16000, 160, 0), (8000, 80, 1), (16000, 80, 0), (8000, 160, 0)])
def test_far_end_bursts_in_session_batches_vs_reference_sessions(fs, frame, with_clean):
    """WebRtcAecmSessions_BufferFarend / _Process (+ the Tick forms for the last far call of a tick) against one reference
    session per stream: every session its own k = 0, 1, 1, 1, 2, 3 far calls per near call, its own 30-frame bursts (jitter
    buffer overflow, replay frames lapped in the device's far ring), its own msInSndCardBuf; host pointers, device
    pointers and the asynchronous form in turn."""
    import torch
    S = 7
    n_calls = 3 * fs // frame if frame == 160 or fs == 8000 else 400
    pats = [call_pattern(20 + 3 * k + frame, n_calls, bursts=True) for k in range(S)]
    k_of = np.stack([p[1] for p in pats], axis=1).astype(np.int64)                     # [n_calls, S]
    k_of[:, 6] = np.minimum(k_of[:, 6], 1)                                             # one session with the common 0 / 1 pattern
    ms_of = np.stack([p[0] for p in pats], axis=1)
    fars = [synth_pair(700 + k, int(k_of[:, k].sum()) * frame // 64 + 1, fs, "mixed")[0] for k in range(S)]
    near = np.stack([synth_pair(700 + k, n_calls * frame // 64 + 1, fs, "mixed")[1][:n_calls * frame] for k in range(S)])
    clean = synth_clean(near) if with_clean else None
    refs = [pyoracle.RefSession(fs, 1, 3) for _ in range(S)]
    sb = aecm.AecmSessions(S, fs, 1, 3)
    cursor = np.zeros(S, dtype=np.int64)
    for i in range(n_calls):
        sl = slice(i * frame, (i + 1) * frame)
        k = k_of[i].copy()
        ms = ms_of[i].copy()
        c = None if clean is None else clean[:, sl]
        style = i % 4
        # the reference side: k far calls, then the near call
        rows = np.zeros((S, max(int(k.max()), 1) * frame), dtype=np.int16)
        want, want_codes = np.empty((S, frame), np.int16), np.empty(S, np.int32)
        for q in range(S):
            rows[q, :k[q] * frame] = fars[q][cursor[q]:cursor[q] + k[q] * frame]
            for j in range(k[q]):
                assert refs[q].buffer_farend(rows[q, j * frame:(j + 1) * frame]) == 0
            want_codes[q], want[q] = refs[q].process(near[q, sl], None if c is None else c[q], int(ms[q]))
            cursor[q] += k[q] * frame
        if style in (0, 1):
            # every far call through BufferFarend (host / device pointers), then Process
            if style == 0:
                assert sb.buffer_farend_host(rows, frame, int(k.max()), k.astype(np.uint8)) == 0
                rc, out, codes = sb.process_host(near[:, sl], clean=c, ms_per_session=ms)
            else:
                drows = torch.from_numpy(rows).cuda()
                dnear = torch.from_numpy(np.ascontiguousarray(near[:, sl])).cuda()
                dclean = None if c is None else torch.from_numpy(np.ascontiguousarray(c)).cuda()
                dout = torch.empty_like(dnear)
                torch.cuda.synchronize()
                assert sb.buffer_farend_device(drows.data_ptr(), rows.shape[1], frame, int(k.max()), k.astype(np.uint8), asynchronous=True) == 0
                rc = sb.process_device(dnear.data_ptr(), dout.data_ptr(), frame, frame, clean_ptr=None if dclean is None else dclean.data_ptr(),
                                       ms_per_session=ms)
                out = dout.cpu().numpy()
                codes = np.where((ms < 0) | (ms > 500), aecm.ffi.AECM_BAD_PARAMETER_WARNING, 0)
            assert rc == next((int(x) for x in want_codes if x), 0)
        else:
            # the last far call of every session rides in the tick (TickFlags: k in {0, 1}), the ones before it in a burst
            extra = np.maximum(k - 1, 0)
            if extra.max() > 0:
                assert sb.buffer_farend_host(rows, frame, int(extra.max()), extra.astype(np.uint8)) == 0
            last = np.zeros((S, frame), dtype=np.int16)
            for q in range(S):
                if k[q]:
                    last[q] = rows[q, (k[q] - 1) * frame:k[q] * frame]
            fl = np.where(k == 0, aecm.ffi.SESSION_NO_FAREND, 0).astype(np.uint8)
            rc, out, codes = sb.tick_host_per_session(last, near[:, sl], ms, c, flags=fl)
        assert np.array_equal(codes, want_codes), (i, style)
        bad = np.nonzero((out != want).any(axis=1))[0]
        assert bad.size == 0, (fs, frame, i, style, bad.tolist())
    # argument validation, in the reference's order (echo_control_mobile.cc:195-213)
    z = np.zeros((S, 3 * frame), np.int16)
    assert sb.lib.WebRtcAecmSessions_BufferFarendHost(sb.h, None, frame, frame, 1, None) == aecm.ffi.AECM_NULL_POINTER_ERROR
    assert sb.buffer_farend_host(z, 100, 1) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    assert sb.buffer_farend_host(z, frame, 4) == aecm.ffi.AECM_BAD_PARAMETER_ERROR                     # rows too short for 4 calls
    assert sb.buffer_farend_host(z, frame, 2, np.full(S, 3, np.uint8)) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    assert sb.buffer_farend_host(z, frame, 0) == 0
    sb.close()


@_needs_ref
@pytest.mark.parametrize("fs,frame", [(16000, 160), (8000, 80)])
def test_far_end_burst_of_255_calls_overflows_the_jitter_buffer(fs, frame):
    """The largest burst the ABI takes (255 WebRtcAecm_BufferFarend calls per session in one launch; host pointers: staged in
    rounds) into a jitter buffer that holds 4 000 samples: the reference truncates the call that no longer fits and drops
    the rest (ring_buffer.c:142-170).  Before and after: ordinary ticks, which must go on from the overflowed buffer exactly
    as the reference's sessions do (delay compensation, read-pointer moves)."""
    S, n_ticks = 3, 90
    pairs = [synth_pair(8100 + k, (n_ticks + 260) * frame // 64 + 1, fs, "mixed") for k in range(S)]
    refs = [pyoracle.RefSession(fs, 1, 3) for _ in range(S)]
    sb = aecm.AecmSessions(S, fs, 1, 3)
    cursor = np.zeros(S, dtype=np.int64)

    def tick(i):
        n = frame
        far = np.stack([pairs[k][0][cursor[k]:cursor[k] + n] for k in range(S)])
        near = np.stack([pairs[k][1][i * frame:(i + 1) * frame] for k in range(S)])
        rc, out = sb.tick_host(far, near, 40)
        for k in range(S):
            assert refs[k].buffer_farend(far[k]) == 0
            rc1, o1 = refs[k].process(near[k], None, 40)
            assert rc == rc1 and np.array_equal(out[k], o1), (fs, i, k)
        cursor[:] += n
    for i in range(30):
        tick(i)
    # 255 calls each (session 1: 254, session 2: none): far more than the buffer holds
    calls = np.array([255, 254, 0], dtype=np.uint8)
    rows = np.stack([pairs[k][0][cursor[k]:cursor[k] + 255 * frame] for k in range(S)])
    assert sb.buffer_farend_host(rows, frame, 255, calls) == 0
    for k in range(S):
        for j in range(int(calls[k])):
            assert refs[k].buffer_farend(rows[k, j * frame:(j + 1) * frame]) == 0
    cursor += calls.astype(np.int64) * frame
    for i in range(30, 60):
        tick(i)
    sb.close()


class _LoggedRef:
    """A reference session that remembers every call it received, so that a second reference session in the same state can be
    made by replaying them (the reference has no way to copy an instance; migration tests need the copy)."""

    def __init__(self, fs):
        self.fs = fs
        self.ref = pyoracle.RefSession(fs, 1, 3)
        self.log = []

    def call(self, name, *args):
        self.log.append((name, args))
        return getattr(self.ref, name)(*args)

    def clone(self):
        c = _LoggedRef(self.fs)
        for name, args in self.log:
            c.call(name, *args)
        return c


@_needs_ref
@pytest.mark.parametrize("seed,fs", [(1, 16000), (2, 8000), (3, 16000)])
def test_session_api_fuzz_vs_reference_sessions(seed, fs):
    """Everything the session-batch ABI offers, interleaved at random for 1 500 steps on two AecmSessions objects of 6 sessions:
    ticks in their three forms (uniform, per-session delay + flags incl. two 80-sample calls in a 160-sample tick, far-end
    bursts + near-end only) with the frame size changing from tick to tick, bursts of 0 .. 6 (sometimes 40) far frames,
    slots re-initialised, re-configured and re-seeded, and live sessions migrated between the two objects.  After every call
    every session must equal a reference session that received exactly the same reference calls."""
    rs = np.random.RandomState(seed)
    S, steps = 6, 1500
    objs = [aecm.AecmSessions(S, fs, 1, 3) for _ in range(2)]
    refs = [[_LoggedRef(fs) for _ in range(S)] for _ in range(2)]
    L = 900 * 160
    audio = [[synth_pair(9000 + 10 * o + k, L // 64 + 1, fs, "mixed") for k in range(S)] for o in range(2)]
    fpos = np.zeros((2, S), dtype=np.int64)                     # far / near read positions per session (wrap inside the recordings)
    npos = np.zeros((2, S), dtype=np.int64)

    def take(o, k, which, n):
        pos = fpos if which == 0 else npos
        a = audio[o][k][which]
        start = int(pos[o, k]) % (a.size - 200 * 160)
        pos[o, k] += n
        return a[start:start + n]

    n_migrations = n_bursts = n_cancelling = 0
    for step in range(steps):
        o = int(rs.randint(0, 2))
        sb, rf = objs[o], refs[o]
        op = rs.rand()
        if op < 0.62:                                             # ---- a tick, in one of its forms
            n = int(rs.choice([80, 160]))
            ms = (40 + rs.randint(-15, 16, size=S)).astype(np.int16)
            if rs.rand() < 0.05:
                ms[rs.randint(0, S)] = rs.choice([-5, 700])
            form = int(rs.randint(0, 3))
            far = np.stack([take(o, k, 0, n) for k in range(S)])
            near = np.stack([take(o, k, 1, n) for k in range(S)])
            want, want_codes = np.empty((S, n), np.int16), np.empty(S, np.int32)
            if form == 0:                                         # uniform: one far + one near call each
                ms[:] = ms[0]
                rc, out = sb.tick_host(far, near, int(ms[0]))
                codes = np.full(S, rc)
                flags = np.zeros(S, np.uint8)
            elif form == 1:                                       # per-session delays and flags
                flags = rs.choice([0, 0, 0, aecm.ffi.SESSION_NO_FAREND, aecm.ffi.SESSION_SPLIT_CALLS if n == 160 else 0], size=S).astype(np.uint8)
                rc, out, codes = sb.tick_host_per_session(far, near, ms, flags=flags)
            else:                                                 # far-end burst of 0 .. 3 frames of this size, then the near-end calls alone
                k_s = rs.randint(0, 4, size=S)
                rows = np.zeros((S, 3 * n), np.int16)
                rows[:, :n] = far
                for k in range(S):
                    for j in range(1, int(k_s[k])):
                        rows[k, j * n:(j + 1) * n] = take(o, k, 0, n)
                assert sb.buffer_farend_host(rows, n, 3, k_s.astype(np.uint8)) == 0
                rc, out, codes = sb.process_host(near, ms_per_session=ms)
                flags = None
            for k in range(S):
                if form == 2:
                    for j in range(int(k_s[k])):
                        assert rf[k].call("buffer_farend", rows[k, j * n:(j + 1) * n].copy()) == 0
                    want_codes[k], want[k] = rf[k].call("process", near[k].copy(), None, int(ms[k]))
                elif flags[k] & aecm.ffi.SESSION_SPLIT_CALLS:
                    c = 0
                    for h in range(2):
                        assert rf[k].call("buffer_farend", far[k, h * 80:(h + 1) * 80].copy()) == 0
                        c1, want[k, h * 80:(h + 1) * 80] = rf[k].call("process", near[k, h * 80:(h + 1) * 80].copy(), None, int(ms[k]))
                        c = c or c1
                    want_codes[k] = c
                else:
                    if not flags[k] & aecm.ffi.SESSION_NO_FAREND:
                        assert rf[k].call("buffer_farend", far[k].copy()) == 0
                    want_codes[k], want[k] = rf[k].call("process", near[k].copy(), None, int(ms[k]))
            assert np.array_equal(codes, want_codes), (seed, step, form)
            bad = np.nonzero((out != want).any(axis=1))[0]
            assert bad.size == 0, (seed, step, form, n, bad.tolist())
            n_cancelling += int((out != near).any(axis=1).sum())                # calls past the start-up copy
        elif op < 0.74:                                           # ---- a far-end burst on its own
            n = int(rs.choice([80, 160]))
            kmax = 40 if rs.rand() < 0.1 else 6
            k_s = rs.randint(0, kmax + 1, size=S)
            rows = np.stack([take(o, k, 0, kmax * n) for k in range(S)])
            assert sb.buffer_farend_host(rows, n, kmax, k_s.astype(np.uint8)) == 0
            for k in range(S):
                for j in range(int(k_s[k])):
                    assert rf[k].call("buffer_farend", rows[k, j * n:(j + 1) * n].copy()) == 0
            n_bursts += 1
        elif op < 0.80:                                           # ---- a slot is recycled
            k = int(rs.randint(0, S))
            assert sb.init_session(k) == 0 and rf[k].call("init", fs) == 0
        elif op < 0.87:
            k, cfg = int(rs.randint(0, S)), (int(rs.randint(0, 2)), int(rs.randint(0, 5)))
            assert sb.set_config_session(k, *cfg) == 0 and rf[k].call("set_config", *cfg) == 0
        elif op < 0.91:
            k, path = int(rs.randint(0, S)), rs.randint(0, 6000, size=65).astype(np.int16)
            assert sb.init_echo_path(k, path) == 0 and rf[k].call("init_echo_path", path.copy()) == 0
        elif op < 0.95:
            k = int(rs.randint(0, S))
            rc, p = sb.get_echo_path(k)
            rc1, p1 = rf[k].call("get_echo_path")
            assert rc == rc1 == 0 and np.array_equal(p, p1), (seed, step)
        else:                                                     # ---- a live session moves to the other object (the source goes on as well)
            k, j = int(rs.randint(0, S)), int(rs.randint(0, S))
            rc, snap = sb.export_session(k)
            assert rc == 0 and objs[1 - o].import_session(j, snap) == 0
            refs[1 - o][j] = rf[k].clone()
            fpos[1 - o, j], npos[1 - o, j] = fpos[o, k], npos[o, k]
            audio[1 - o][j] = audio[1 - o][j]                     # (the imported session continues on ITS slot's audio: any audio is a valid continuation)
            n_migrations += 1
    print(f"fuzz seed {seed}: {n_migrations} migrations, {n_bursts} bursts, {n_cancelling} session-calls past start-up")
    assert n_migrations >= 40 and n_bursts >= 100 and n_cancelling >= 2000
    for sb in objs:
        sb.close()


@_needs_ref
def test_session_churn_slots_recycled_mid_run():
    """A media server recycling slots: sessions are re-initialised, re-configured and re-seeded (echo path) by index
    while the others keep running, with per-session delays and underruns; every slot must equal a reference
    session that received exactly the same calls (WebRtcAecm_Init / set_config / InitEchoPath / BufferFarend / Process)."""
    rs = np.random.RandomState(77)
    for fs, frame in ((16000, 160), (8000, 80)):
        S = 8
        n_calls = 3 * fs // frame
        pairs = [synth_pair(900 + k, n_calls * frame // 64 + 1, fs, "mixed") for k in range(S)]
        far = np.stack([p[0][:n_calls * frame] for p in pairs])
        near = np.stack([p[1][:n_calls * frame] for p in pairs])
        refs = [pyoracle.RefSession(fs, 1, 3) for _ in range(S)]
        sb = aecm.AecmSessions(S, fs, 1, 3)
        base_ms = np.array([40, 40, 60, 40, 100, 40, 40, 25], dtype=np.int16)
        path = (np.arange(65) * 53 % 3000 + 100).astype(np.int16)
        for i in range(n_calls):
            sl = slice(i * frame, (i + 1) * frame)
            # churn events between ticks
            if i in (n_calls // 4, n_calls // 2):
                for k in (1, 5) if i == n_calls // 4 else (1, 2, 6):
                    assert sb.init_session(k) == 0 and refs[k].init(fs) == 0
                    cfg = (int(rs.randint(0, 2)), int(rs.randint(0, 5)))
                    assert sb.set_config_session(k, *cfg) == 0 and refs[k].set_config(*cfg) == 0
            if i == n_calls // 3:
                assert sb.init_echo_path(3, path) == 0 and refs[3].init_echo_path(path) == 0
                assert sb.set_config_session(4, 0, 1) == 0 and refs[4].set_config(0, 1) == 0
            ms = base_ms.copy()
            fl = np.zeros(S, dtype=np.uint8)
            if i % 37 == 36:
                fl[[0, 5]] = aecm.ffi.SESSION_NO_FAREND
            rc, out, codes = sb.tick_host_per_session(far[:, sl], near[:, sl], ms, flags=fl)
            for k in range(S):
                if not fl[k]:
                    assert refs[k].buffer_farend(far[k, sl]) == 0
                rc1, o1 = refs[k].process(near[k, sl], None, int(ms[k]))
                assert codes[k] == rc1 and np.array_equal(out[k], o1), (fs, i, k)
        rc3, p3 = sb.get_echo_path(3)
        rc3r, p3r = refs[3].get_echo_path()
        assert rc3 == rc3r == 0 and np.array_equal(p3, p3r)
        assert sb.init_session(S) == aecm.ffi.AECM_BAD_PARAMETER_ERROR and sb.set_config_session(0, 1, 7) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
        sb.close()


@_needs_ref
@pytest.mark.parametrize("fs", [16000, 8000])
def test_replay_frame_outlives_its_place_in_the_far_ring(fs):
    """farendOld[1] (the frame an underrun of the second 80-sample frame of a call replays) goes unused while a session
    makes two 80-sample calls per tick (those only ever use slot 0); on the device that frame is a position in the far
    ring, which the ring's write position laps after 8 192 accepted samples -- it has to be moved to its replay row in
    time.  Then a run of underruns replays both slots.  Against reference sessions, tick by tick.  (The host-planned tick
    forms of early round 2 could not express this call pattern: their sample tags had to stay inside the far ring.)"""
    frame, S, n_ticks = 160, 3, 420
    pairs = [synth_pair(5150 + k, n_ticks * frame // 64 + 1, fs, "mixed") for k in range(S)]
    far = np.stack([p[0][:n_ticks * frame] for p in pairs])
    near = np.stack([p[1][:n_ticks * frame] for p in pairs])
    refs = [pyoracle.RefSession(fs, 1, 3) for _ in range(S)]
    sb = aecm.AecmSessions(S, fs, 1, 3)
    for i in range(n_ticks):
        sl = slice(i * frame, (i + 1) * frame)
        ph = i % 200
        fl = np.zeros(S, dtype=np.uint8)
        if 60 <= ph < 150:
            fl[:2] |= aecm.ffi.SESSION_SPLIT_CALLS                       # session 2 keeps making one 160-sample call
        if 138 <= ph < 160:
            fl |= aecm.ffi.SESSION_NO_FAREND
        ms = np.array([40, 38 + ph % 5, 40], dtype=np.int16)
        rc, out, codes = sb.tick_host_per_session(far[:, sl], near[:, sl], ms, flags=fl)
        assert rc == 0
        for k in range(S):
            halves = ((0, 80), (80, 160)) if fl[k] & aecm.ffi.SESSION_SPLIT_CALLS else ((0, 160),)
            for a, b in halves:
                if not fl[k] & aecm.ffi.SESSION_NO_FAREND:
                    assert refs[k].buffer_farend(far[k, sl][a:b]) == 0
                rc1, o1 = refs[k].process(near[k, sl][a:b], None, int(ms[k]))
                assert rc1 == 0 and np.array_equal(out[k, a:b], o1), (fs, i, k, a)
    sb.close()


@_needs_ref
@pytest.mark.parametrize("fs,frame", [(16000, 160), (8000, 80), (8000, 160)])
def test_65536_live_sessions_property(fs, frame):
    """The serving claim at its full size, through a size-independent property: 65 536 sessions replicate 32 distinct
    (audio, msInSndCardBuf) pairs, so every session must equal -- tick by tick, over start-up, delay compensation and
    steady state -- the reference session of the pair it replicates (device-resident audio, per-session delays)."""
    import torch
    S, U, n_ticks = 65536, 32, 48
    pairs = [synth_pair(7000 + k, n_ticks * frame // 64 + 1, fs, "mixed") for k in range(U)]
    far_u = np.stack([p[0][:n_ticks * frame] for p in pairs])
    near_u = np.stack([p[1][:n_ticks * frame] for p in pairs])
    ms_u = (25 + 4 * np.arange(U)).astype(np.int16)                       # 25 .. 149 ms: different start-up lengths and stuffing
    refs = [pyoracle.RefSession(fs, 1, 3) for _ in range(U)]
    exp = np.empty((U, n_ticks * frame), dtype=np.int16)
    for i in range(n_ticks):
        sl = slice(i * frame, (i + 1) * frame)
        for k in range(U):
            assert refs[k].buffer_farend(far_u[k, sl]) == 0
            rc, exp[k, sl] = refs[k].process(near_u[k, sl], None, int(ms_u[k]))
            assert rc == 0
    idx = torch.arange(S) % U
    dfar = torch.from_numpy(far_u)[idx].contiguous().cuda()
    dnear = torch.from_numpy(near_u)[idx].contiguous().cuda()
    dexp = torch.from_numpy(exp).cuda()
    ms = ms_u[idx.numpy()]
    # Tick() uses ONE row stride for far, near and out: give the output the recordings' stride too
    dout = torch.empty_like(dnear)
    torch.cuda.synchronize()
    sb = aecm.AecmSessions(S, fs, 1, 3)
    for i in range(n_ticks):
        off = i * frame * 2                                                # bytes into every row
        assert sb.tick_device_per_session(dfar.data_ptr() + off, dnear.data_ptr() + off, dout.data_ptr() + off, dfar.shape[1], frame, ms) == 0
    torch.cuda.synchronize()
    sb.close()
    assert torch.equal(dout.view(S // U, U, -1), dexp.unsqueeze(0).expand(S // U, U, -1))


@_needs_ref
@pytest.mark.parametrize("fs,frame", [(16000, 160), (8000, 80)])
def test_65536_live_sessions_with_far_end_bursts_property(fs, frame):
    """Far-end bursts at the full serving size: 65 536 sessions replicate 32 distinct (audio, burst pattern, msInSndCardBuf) triples
    -- k = 0, 1, 1, 1, 2, 3 far calls per near call and 30-frame bursts that overflow the jitter buffer, through
    WebRtcAecmSessions_BufferFarend + _Process on device pointers -- and every session must equal the reference session of the
    triple it replicates, call by call."""
    import torch
    S, U, n_calls = 65536, 32, 110
    pats = [call_pattern(50 + k, n_calls, bursts=True) for k in range(U)]
    k_u = np.stack([p[1] for p in pats], axis=1).astype(np.int64)                # [n_calls, U]
    ms_u = np.stack([p[0] for p in pats], axis=1)
    fars = [synth_pair(7100 + k, int(k_u[:, k].sum()) * frame // 64 + 1, fs, "mixed")[0] for k in range(U)]
    near_u = np.stack([synth_pair(7100 + k, n_calls * frame // 64 + 1, fs, "mixed")[1][:n_calls * frame] for k in range(U)])
    refs = [pyoracle.RefSession(fs, 1, 3) for _ in range(U)]
    idx = torch.arange(S) % U
    idx_np = idx.numpy()
    dnear_all = torch.from_numpy(near_u)[idx].contiguous().cuda()
    dout = torch.empty((S, frame), dtype=torch.int16, device="cuda")
    sb = aecm.AecmSessions(S, fs, 1, 3)
    cursor = np.zeros(U, dtype=np.int64)
    for i in range(n_calls):
        sl = slice(i * frame, (i + 1) * frame)
        k = k_u[i]
        kmax = int(k.max())
        rows = np.zeros((U, max(kmax, 1) * frame), dtype=np.int16)
        want = np.empty((U, frame), dtype=np.int16)
        codes = np.empty(U, dtype=np.int32)
        for q in range(U):
            rows[q, :k[q] * frame] = fars[q][cursor[q]:cursor[q] + k[q] * frame]
            for j in range(k[q]):
                assert refs[q].buffer_farend(rows[q, j * frame:(j + 1) * frame]) == 0
            codes[q], want[q] = refs[q].process(near_u[q, sl], None, int(ms_u[i, q]))
            cursor[q] += k[q] * frame
        if kmax:
            drows = torch.from_numpy(rows)[idx].contiguous().cuda()
            torch.cuda.synchronize()
            assert sb.buffer_farend_device(drows.data_ptr(), rows.shape[1], frame, kmax, k[idx_np].astype(np.uint8)) == 0
        dn = dnear_all[:, sl].contiguous()
        torch.cuda.synchronize()
        rc = sb.process_device(dn.data_ptr(), dout.data_ptr(), frame, frame, ms_per_session=ms_u[i][idx_np])
        assert rc == next((int(c) for c in codes if c), 0), i
        assert torch.equal(dout.view(S // U, U, frame), torch.from_numpy(want).cuda().unsqueeze(0).expand(S // U, U, frame)), i
    sb.close()


@_needs_ref
def test_sixteen_thousand_sessions_each_with_its_own_history():
    """The serving shape at scale: 16 384 live sessions, EVERY one with its own msInSndCardBuf walk, its own far-end
    underruns, its own call shape (one 160-sample call or two of 80) and its own age (slots re-initialised at random
    ticks) -- more distinct histories than any host-side bookkeeping could follow.  A sample of sessions is compared,
    tick by tick, with reference sessions that received exactly the same calls."""
    fs, frame, S, n_ticks = 16000, 160, 16384, 150
    rs = np.random.RandomState(2024)
    base_far, base_near = synth_pair(4242, (n_ticks * frame + 4096) // 64 + 1, fs, "mixed")
    shift = rs.randint(0, 4096, size=S)                                  # every session hears its own cut of the recording
    # AECM_SOAK_WATCH=n: compare n sessions instead of 24 (a one-off soak after changes to the tick kernel's build: 4 096
    # watched sessions take about two minutes of host time)
    n_watch = max(3, int(os.environ.get("AECM_SOAK_WATCH", "24")))
    watch = sorted(set([0, 1, S - 1] + list(rs.randint(0, S, size=n_watch - 3))))
    refs = {k: pyoracle.RefSession(fs, 1, 3) for k in watch}
    sb = aecm.AecmSessions(S, fs, 1, 3)
    ms = (20 + rs.randint(0, 80, size=S)).astype(np.int32)
    idx = np.arange(frame)[None, :]
    for i in range(n_ticks):
        pos = (shift + i * frame)[:, None] + idx
        far, near = base_far[pos], base_near[pos]
        ms = np.clip(ms + rs.randint(-7, 8, size=S), -20, 560)           # a walk per session, now and then out of range
        fl = np.where(rs.rand(S) < 0.04, aecm.ffi.SESSION_NO_FAREND, 0).astype(np.uint8)
        fl |= np.where(rs.rand(S) < 0.3, aecm.ffi.SESSION_SPLIT_CALLS, 0).astype(np.uint8)
        if i and i % 23 == 0:                                             # churn: a few slots start over (some of them watched)
            for k in list(rs.randint(0, S, size=6)) + [watch[(i // 23) % len(watch)]]:
                assert sb.init_session(int(k)) == 0
                if k in refs:
                    assert refs[k].init(fs) == 0 and refs[k].set_config(1, 3) == 0
        rc, out, codes = sb.tick_host_per_session(far, near, ms.astype(np.int16), flags=fl)
        assert rc in (0, aecm.ffi.AECM_BAD_PARAMETER_WARNING)
        for k in watch:
            halves = ((0, 80), (80, 160)) if fl[k] & aecm.ffi.SESSION_SPLIT_CALLS else ((0, 160),)
            rc_ref = 0
            for a, b in halves:
                if not fl[k] & aecm.ffi.SESSION_NO_FAREND:
                    assert refs[k].buffer_farend(far[k, a:b]) == 0
                rc1, o1 = refs[k].process(near[k, a:b], None, int(ms[k]))
                rc_ref = rc_ref or rc1
                assert np.array_equal(out[k, a:b], o1), (i, k, a)
            assert codes[k] == rc_ref, (i, k)
    sb.close()


def test_tick_argument_validation():
    """nrOfSamples is validated as a size_t (2^32 + 80 is not 80), strides must cover the tick."""
    import ctypes as C
    lib = aecm.load()
    sb = aecm.AecmSessions(2, 16000)
    z = np.zeros((2, 160), np.int16)
    o = np.zeros_like(z)
    args = (z.ctypes.data, z.ctypes.data, None, o.ctypes.data)
    assert lib.WebRtcAecmSessions_TickHost(sb.h, *args, 160, C.c_size_t(2 ** 32 + 80), 40) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    assert lib.WebRtcAecmSessions_TickHost(sb.h, *args, 160, 100, 40) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    assert lib.WebRtcAecmSessions_TickHost(sb.h, *args, 100, 160, 40) == aecm.ffi.AECM_BAD_PARAMETER_ERROR      # stride < n
    assert lib.WebRtcAecmSessions_TickHost(sb.h, *args, 160, 160, 40) == 0
    with pytest.raises(ValueError):
        sb.tick_host(np.zeros((3, 160), np.int16), np.zeros((3, 160), np.int16))
    with pytest.raises(ValueError):
        sb.tick_host(z, np.zeros((2, 80), np.int16))
    sb.close()


def test_config4_8khz_32768_streams_property():
    """BASELINE config 4 at full size (8 kHz mode, 32768 streams) and past start-up (1 100 blocks: all three start-up
    regimes): replicated streams must equal the oracle's answer for the 64 distinct seeds they replicate
    (size-independent property, as for config 3)."""
    _replicated_full_size_run(32768, 1100, 8000, 64, 1400)


def test_recordings_batch_larger_than_the_scratch_budget():
    """ProcessRecordings cuts the streams into chunks that fit its fixed scratch budget and folds the stream index
    into grid.x: a batch of more than 65535 recordings (the old grid.y limit) must work and equal a small batch."""
    import torch
    S, fs, frame, n_calls, U = 66000, 16000, 160, 150, 4
    pairs = [synth_pair(1700 + k, n_calls * frame // 64 + 1, fs, "mixed") for k in range(U)]
    far = np.stack([p[0][:n_calls * frame] for p in pairs])
    near = np.stack([p[1][:n_calls * frame] for p in pairs])
    small = aecm.AecmBatch(U, fs, 1, 1)
    rc, exp = small.process_recordings_host(far, near, frame, 40)
    assert rc == 0
    idx = torch.arange(S) % U
    dfar = torch.from_numpy(far)[idx].contiguous().cuda()
    dnear = torch.from_numpy(near)[idx].contiguous().cuda()
    dout = torch.zeros_like(dnear)
    torch.cuda.synchronize()
    big = aecm.AecmBatch(S, fs, 1, 1)
    assert big.process_recordings_device(dfar.data_ptr(), dnear.data_ptr(), dout.data_ptr(), n_calls * frame, frame, n_calls, 40) == 0
    assert torch.equal(dout.cpu(), torch.from_numpy(exp)[idx])


def test_device_precondition_audit_build(tmp_path):
    """Every "provably fits" claim of the block kernel (24-bit multiplies, int16 narrowings dropped as the identity;
    aecm_ops.h mul24 / as_i16) verified ON THE DEVICE: libaecm_mi355x_checked.so (same sources, -DAECM_CHECKED) counts
    violations while the adversarial corpus, the mixed-profile streams of both rates, the clean-input path, random
    full-range echo paths and states restored through ImportState from long runs go through it.  Counters must stay 0
    and the outputs must still equal the oracle's.  The shipped library has no such counters (12001)."""
    import os
    import subprocess
    import sys
    import textwrap
    from webrtc_aecm_amd import build
    with pytest.raises(aecm.AecmError) as e:
        aecm.check_counters(0)
    assert e.value.code == aecm.ffi.AECM_UNSUPPORTED_FUNCTION_ERROR
    assert build.LIB_CHECKED.exists()
    script = tmp_path / "audit.py"
    script.write_text(textwrap.dedent("""
        import sys
        import numpy as np
        import webrtc_aecm_amd as aecm
        from helpers import adversarial_cases, synth_streams, oracle_batch, stream_config
        from oracle import pyoracle
        from webrtc_aecm_amd.synth import synth_clean
        assert aecm.check_counters(0, reset=True) is not None
        for fs in (16000, 8000):
            sel = [c for c in adversarial_cases(n_cases=36, n_blocks=700, seed=19) if c["fs"] == fs]
            b = aecm.AecmBatch(len(sel), fs)
            for k, c in enumerate(sel):
                b.set_config(c["cng"], c["echo_mode"], k, 1)
                if c["path"] is not None:
                    b.init_echo_path(k, c["path"])
            out = b.process_host(np.stack([c["far"] for c in sel]), np.stack([c["near"] for c in sel]))
            for k, c in enumerate(sel):
                o = pyoracle.OracleStream(fs, c["cng"], c["echo_mode"])
                if c["path"] is not None:
                    o.init_echo_path(c["path"])
                assert np.array_equal(out[k], o.process(c["far"], c["near"])), (fs, k)
            S, T = 32, 2100
            seeds = list(range(5000, 5000 + S))
            cfgs = [stream_config(s) for s in range(S)]
            far, near = synth_streams(seeds, T, fs)
            b = aecm.AecmBatch(S, fs)
            for s, (cng, em) in enumerate(cfgs):
                b.set_config(cng, em, s, 1)
            half = (T // 2) * 64
            out1 = b.process_host(far[:, :half], near[:, :half])
            # states restored through ImportState from a long run continue in another batch
            b2 = aecm.AecmBatch(S, fs)
            for s in range(S):
                b2.import_state(s, b.export_state(S - 1 - s))
            out2 = b2.process_host(far[::-1, half:].copy(), near[::-1, half:].copy())[::-1]
            exp, _ = oracle_batch(seeds, T, fs, cfgs)
            assert np.array_equal(np.concatenate([out1, out2], axis=1), exp), fs
            clean = synth_clean(near[:8, :600 * 64])
            bc = aecm.AecmBatch(8, fs)
            bc.process_host(far[:8, :600 * 64], near[:8, :600 * 64], clean)
        # full-scale white noise on both ends, loud channel, every echo mode
        rs = np.random.RandomState(3)
        far = rs.randint(-32768, 32768, size=(10, 500 * 64)).astype(np.int16)
        near = rs.randint(-32768, 32768, size=(10, 500 * 64)).astype(np.int16)
        b = aecm.AecmBatch(10, 16000)
        for k in range(10):
            b.set_config(k % 2, k % 5, k, 1)
            b.init_echo_path(k, np.full(65, 32767 if k % 3 else -32768, np.int16))
        b.process_host(far, near)
        c = aecm.check_counters(0)
        print("AUDIT", int(c[0]), int(c[1]))
    """))
    env = dict(os.environ, AECM_LIB_PATH=str(build.LIB_CHECKED),
               PYTHONPATH=os.pathsep.join([str(GOLDEN.parent.parent), str(GOLDEN.parent), os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "AUDIT 0 0" in r.stdout, r.stdout[-500:]


@_needs_ref
def test_mixed_call_cadence_in_one_session_batch():
    """Sessions with different call sizes in ONE streaming object (reference: every instance may use 80- or 160-sample
    calls, echo_control_mobile.cc:268,282): on 160-sample ticks some sessions make two 80-sample BufferFarend + Process
    pairs (AECM_SESSION_SPLIT_CALLS), the others one 160-sample pair, with per-session delays, underruns and a
    saturating delay report; every session equals a reference session that received exactly those calls."""
    for fs in (16000, 8000):
        S, n = 8, 160
        n_ticks = 3 * fs // n
        pairs = [synth_pair(1200 + k, n_ticks * n // 64 + 1, fs, "mixed") for k in range(S)]
        far = np.stack([p[0][:n_ticks * n] for p in pairs])
        near = np.stack([p[1][:n_ticks * n] for p in pairs])
        refs = [pyoracle.RefSession(fs, 1, 2) for _ in range(S)]
        sb = aecm.AecmSessions(S, fs, 1, 2)
        split = np.array([0, 1, 0, 1, 1, 0, 1, 0], dtype=bool)
        base_ms = np.array([40, 40, 60, 90, 500, 500, 25, 40], dtype=np.int16)     # 500 ms: the jitter buffer saturates and drops
        for i in range(n_ticks):
            sl = slice(i * n, (i + 1) * n)
            fl = np.where(split, aecm.ffi.SESSION_SPLIT_CALLS, 0).astype(np.uint8)
            if i % 41 == 40:
                fl[[1, 2]] |= aecm.ffi.SESSION_NO_FAREND
            if i == n_ticks // 2:                                  # cadence changes mid-run for two sessions
                split[0], split[1] = True, False
            rc, out, codes = sb.tick_host_per_session(far[:, sl], near[:, sl], base_ms, flags=fl)
            for k in range(S):
                if fl[k] & aecm.ffi.SESSION_SPLIT_CALLS:
                    first = 0
                    for h in (0, 80):
                        if not fl[k] & aecm.ffi.SESSION_NO_FAREND:
                            assert refs[k].buffer_farend(far[k, sl][h:h + 80]) == 0
                        rc1, o1 = refs[k].process(near[k, sl][h:h + 80], None, int(base_ms[k]))
                        first = first or rc1
                        assert np.array_equal(out[k, h:h + 80], o1), (fs, i, k, h)
                    assert codes[k] == first, (fs, i, k)
                else:
                    if not fl[k] & aecm.ffi.SESSION_NO_FAREND:
                        assert refs[k].buffer_farend(far[k, sl]) == 0
                    rc1, o1 = refs[k].process(near[k, sl], None, int(base_ms[k]))
                    assert codes[k] == rc1 and np.array_equal(out[k], o1), (fs, i, k)
        sb.close()
    sb = aecm.AecmSessions(2, 16000)
    z = np.zeros((2, 80), np.int16)
    rc, _, _ = sb.tick_host_per_session(z, z, np.array([40, 40], np.int16), flags=np.array([aecm.ffi.SESSION_SPLIT_CALLS, 0], np.uint8))
    assert rc == aecm.ffi.AECM_BAD_PARAMETER_ERROR                  # two 80-sample calls need a 160-sample tick
    sb.close()


def test_state_snapshot_value_ranges_are_validated():
    """ImportState also refuses values outside the ranges the kernel's arithmetic shortcuts rely on (as_nonneg(supGain),
    int16 members whose narrowing casts are dropped as the identity, Q domains used as shift counts, counters): a blob
    no run of the algorithm can produce must not be run on (aecm_host_state.cpp: ValidateStateImage)."""
    import struct
    fs = 16000
    far, near = synth_streams([9], 300, fs)
    b = aecm.AecmBatch(1, fs)
    b.process_host(far, near)
    blob = b.export_state(0)
    lib = aecm.load()
    dig = b.digest(0)
    scal0 = 32 + 12 * 64 * 4
    S = {"SUPGAIN": 21, "SUPGAIN_OLD": 22, "FARLOG": 8, "FE_MIN": 9, "CURVAD": 14, "SEED": 1, "NLP": 31, "SG_A": 33,
         "B64_NEARFILT": 41, "B64_NOISE": 42, "B64_LOWCTR": 43, "MIN_PROB": 26, "FIXED_DELAY": 32}

    def imp(x):
        return lib.WebRtcAecmBatch_ImportState(b.h, 0, bytes(x), len(x))
    assert imp(blob) == 0
    for name, value in (("SUPGAIN", -1), ("SUPGAIN", 40000), ("SUPGAIN_OLD", 70000), ("FARLOG", 32768), ("FE_MIN", -32769),
                        ("CURVAD", 2), ("SEED", -5), ("NLP", 1 << 20), ("SG_A", -40000), ("B64_NEARFILT", 65536),
                        ("B64_NOISE", -1), ("B64_LOWCTR", 9), ("MIN_PROB", -1), ("FIXED_DELAY", -40000)):
        bad = bytearray(blob)
        struct.pack_into("<i", bad, scal0 + 4 * S[name], value)
        assert imp(bad) == aecm.ffi.AECM_BAD_PARAMETER_ERROR, (name, value)
    # lane-vector words: a far-spectrum Q domain of 20 (5-bit field of the nearFilt word), a negative noise estimate
    v_nearfilt, v_noise = 32 + 5 * 64 * 4, 32 + 6 * 64 * 4
    bad = bytearray(blob)
    w, = struct.unpack_from("<I", bad, v_nearfilt + 4 * 7)
    struct.pack_into("<I", bad, v_nearfilt + 4 * 7, (w & ~(31 << 22)) | (20 << 22))
    assert imp(bad) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    bad = bytearray(blob)
    struct.pack_into("<i", bad, v_noise + 4 * 63, -7)
    assert imp(bad) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    assert np.array_equal(b.digest(0), dig)
    # Control narrows its arguments to int16 like the reference: anything outside [-32768, 100) is refused, not wrapped
    assert lib.WebRtcAecmBatch_Control(b.h, -65436, 1, 0, -1) == aecm.ffi.AECM_BAD_PARAMETER_ERROR     # would wrap to +100
    assert lib.WebRtcAecmBatch_Control(b.h, 99, 1 << 17, 0, -1) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    assert lib.WebRtcAecmBatch_Control(b.h, -1, 1, 0, -1) == 0


@_needs_ref
def test_process_in_place_out_aliases_the_near_end():
    """The reference lets `out` alias nearendNoisy or nearendClean (echo_control_mobile.cc:285-291 copies only when they
    differ; its own main.cc processes in place through a copy).  WebRtcAecm_Process with out == nearendNoisy and with
    out == nearendClean, during start-up and in the steady state, 80- and 160-sample calls, both rates, against the
    reference driven in exactly the same aliased way."""
    lib, ref = aecm.load(), pyoracle.ref_lib()
    for k, (fs, frame, alias) in enumerate(((16000, 160, "noisy"), (16000, 80, "noisy"), (8000, 80, "clean"), (16000, 160, "clean"),
                                            (8000, 160, "noisy"))):
        far, near = synth_pair(2300 + k, 3 * fs // 64, fs, "mixed")
        clean = synth_clean(near)
        n_calls = far.size // frame
        ms_seq, far_present = call_pattern(70 + k, n_calls)
        r = pyoracle.RefSession(fs, 1, 3)
        s = aecm.Aecm()
        assert s.init(fs) == 0 and s.set_config(1, 3) == 0
        for i in range(n_calls):
            sl = slice(i * frame, (i + 1) * frame)
            if far_present[i]:
                f = np.ascontiguousarray(far[sl])
                assert lib.WebRtcAecm_BufferFarend(s.h, f.ctypes.data, frame) == ref.WebRtcAecm_BufferFarend(r.h, f.ctypes.data, frame) == 0
            bufs = []
            for handle, l in ((s.h, lib), (r.h, ref)):
                noisy, cl = near[sl].copy(), clean[sl].copy()
                if alias == "noisy":          # out == nearendNoisy, no clean input
                    rc = l.WebRtcAecm_Process(handle, noisy.ctypes.data, None, noisy.ctypes.data, frame, int(ms_seq[i]))
                    bufs.append((rc, noisy, cl))
                else:                         # out == nearendClean
                    rc = l.WebRtcAecm_Process(handle, noisy.ctypes.data, cl.ctypes.data, cl.ctypes.data, frame, int(ms_seq[i]))
                    bufs.append((rc, cl, noisy))
            (rc_g, out_g, other_g), (rc_r, out_r, other_r) = bufs
            assert rc_g == rc_r, (fs, frame, alias, i)
            assert np.array_equal(out_g, out_r), (fs, frame, alias, i)
            assert np.array_equal(other_g, other_r), "the non-aliased input must be left alone"
        s.close()


@_needs_ref
def test_independent_instances_driven_from_their_own_threads():
    """The reference's threading contract: distinct instances share nothing and may be driven from different
    threads.  Eight host threads each own one WebRtcAecm_* session, two more each own one AecmBatch (4 streams) on the
    same device; every thread's results must equal the reference's for its own inputs."""
    import threading
    fs, frame = 16000, 160
    n_sess, n_calls = 8, 150
    results, errors = {}, []

    def session_worker(t):
        try:
            far, near = synth_pair(3100 + t, n_calls * frame // 64 + 1, fs, "mixed")
            ms_seq, far_present = call_pattern(300 + t, n_calls)
            s = aecm.Aecm()
            assert s.init(fs) == 0 and s.set_config(1, t % 5) == 0
            results[("s", t)] = drive_session(s, far, near, frame, ms_seq, far_present)
            s.close()
        except Exception as e:                       # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))

    def batch_worker(t):
        try:
            seeds = [3200 + 10 * t + k for k in range(4)]
            far, near = synth_streams(seeds, 400, fs)
            b = aecm.AecmBatch(4, fs, 1, 2)
            outs = [b.process_host(far[:, c * 6400:(c + 1) * 6400], near[:, c * 6400:(c + 1) * 6400]) for c in range(4)]
            results[("b", t)] = (np.concatenate(outs, axis=1), [b.digest(k) for k in range(4)])
            b.close()
        except Exception as e:                       # noqa: BLE001
            errors.append((t, repr(e)))
    threads = [threading.Thread(target=session_worker, args=(t,)) for t in range(n_sess)] + \
              [threading.Thread(target=batch_worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(600)
    assert not errors, errors
    for t in range(n_sess):
        far, near = synth_pair(3100 + t, n_calls * frame // 64 + 1, fs, "mixed")
        ms_seq, far_present = call_pattern(300 + t, n_calls)
        exp, exp_codes = drive_session(pyoracle.RefSession(fs, 1, t % 5), far, near, frame, ms_seq, far_present)
        got, codes = results[("s", t)]
        assert np.array_equal(codes, exp_codes) and np.array_equal(got, exp), t
    for t in range(2):
        seeds = [3200 + 10 * t + k for k in range(4)]
        far, near = synth_streams(seeds, 400, fs)
        out, digs = results[("b", t)]
        for k in range(4):
            r = pyoracle.RefCoreStream(fs, 1, 2)
            assert np.array_equal(out[k], r.process(far[k], near[k])) and np.array_equal(digs[k], r.digest()), (t, k)


def test_c_abi_under_ubsan():
    """The C-ABI tests once more on libaecm_mi355x_ubsan.so: the same gfx950 kernels under host objects (engine, sessions,
    C API, session flow, schedule, host state) compiled with -fsanitize=undefined -fno-sanitize-recover.  Any
    report aborts the child process."""
    import os
    import subprocess
    import sys
    from webrtc_aecm_amd import build
    try:                                   # opt-in test infrastructure: prebuilt by __graft_entry__.build(), (re)built here if stale
        build.build_ubsan()
    except RuntimeError as e:
        pytest.skip(f"no sanitizer twin of the library: {e}")
    env = dict(os.environ, AECM_LIB_PATH=str(build.LIB_UBSAN), UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    sel = ("session_abi or snapshot or echo_path or control or tick_major or ragged or chunked or clean_input or recordings_equal or "
           "streaming_session_batch or unaligned or per_session_sound or churn or mixed_call or tick_argument or in_place or cli_single or "
           "chunk_queue or pipelined_launch")
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-x", "-q", "-p", "no:cacheprovider", "-k", sel],
                       env=env, capture_output=True, text=True, timeout=1500)
    log = r.stdout + r.stderr
    if os.environ.get("AECM_SANITIZER_LOG"):
        open(os.environ["AECM_SANITIZER_LOG"], "w").write(log)
    assert r.returncode == 0 and "runtime error" not in log, log[-4000:]
    assert " passed" in log


@_needs_ref
def test_async_ticks_and_registered_host_audio_equal_reference_sessions():
    """WebRtcAecmSessions_TickAsync: ticks enqueued back to back (per-session delays and flags change every tick, so both
    argument slots are reused while ticks are still in flight), one synchronisation at the end; every session's output of
    every tick must equal a reference session driven with the same calls.  Then the same through caller-owned host
    buffers registered once (WebRtcAecmBatch_RegisterHostBuffer): the kernels read and write them in place."""
    import torch
    S, fs, frame, n_calls = 6, 16000, 160, 120
    pairs = [synth_pair(4000 + k, n_calls * frame // 64 + 1, fs, "mixed") for k in range(S)]
    far = np.stack([p[0][:n_calls * frame] for p in pairs])
    near = np.stack([p[1][:n_calls * frame] for p in pairs])
    pats = [call_pattern(400 + k, n_calls) for k in range(S)]
    ms = np.stack([p[0] for p in pats], axis=1)                      # [n_calls][S]
    present = np.stack([p[1] for p in pats], axis=1)
    exp = np.stack([drive_session(pyoracle.RefSession(fs, 1, 3), far[k], near[k], frame, ms[:, k], present[:, k])[0] for k in range(S)])
    exp_codes = np.stack([drive_session(pyoracle.RefSession(fs, 1, 3), far[k], near[k], frame, ms[:, k], present[:, k])[1] for k in range(S)], axis=1)

    # (a) device-resident audio, every tick reads its own slice and writes its own slice of one big output buffer
    dfar, dnear = torch.from_numpy(far).cuda(), torch.from_numpy(near).cuda()
    dout = torch.zeros_like(dnear)
    sb = aecm.AecmSessions(S, fs, 1, 3)
    lib = aecm.load()
    codes = np.zeros((n_calls, S), dtype=np.int32)
    for i in range(n_calls):
        off = i * frame * 2
        fl = np.where(present[i] != 0, 0, aecm.ffi.SESSION_NO_FAREND).astype(np.uint8)
        m = np.ascontiguousarray(ms[i])
        rc = lib.WebRtcAecmSessions_TickAsync(sb.h, dfar.data_ptr() + off, dnear.data_ptr() + off, None, dout.data_ptr() + off, far.shape[1],
                                              frame, 0, m.ctypes.data, fl.ctypes.data, codes[i].ctypes.data, None, None)
        assert rc in (0, aecm.ffi.AECM_BAD_PARAMETER_WARNING), (i, rc)
    assert sb.synchronize() == 0
    assert np.array_equal(codes, exp_codes)
    assert np.array_equal(dout.cpu().numpy(), exp)
    sb.close()

    # (b) the same calls through registered host buffers (zero copy), synchronous and asynchronous ticks alternating
    hfar, hnear = far.copy(), near.copy()
    hout = np.zeros_like(hnear)
    pf, pn, po = (aecm.register_host_buffer(a) for a in (hfar, hnear, hout))
    sb = aecm.AecmSessions(S, fs, 1, 3)
    for i in range(n_calls):
        off = i * frame * 2
        fl = np.where(present[i] != 0, 0, aecm.ffi.SESSION_NO_FAREND).astype(np.uint8)
        m = np.ascontiguousarray(ms[i])
        if i % 3 == 0:
            rc = lib.WebRtcAecmSessions_TickFlags(sb.h, pf + off, pn + off, None, po + off, far.shape[1], frame, m.ctypes.data, fl.ctypes.data, None)
        else:
            rc = lib.WebRtcAecmSessions_TickAsync(sb.h, pf + off, pn + off, None, po + off, far.shape[1], frame, 0, m.ctypes.data, fl.ctypes.data,
                                                  None, None, None)
        assert rc in (0, aecm.ffi.AECM_BAD_PARAMETER_WARNING), (i, rc)
    assert sb.synchronize() == 0
    assert np.array_equal(hout, exp)
    sb.close()
    for a in (hfar, hnear, hout):
        aecm.unregister_host_buffer(a)
    # a block batch on registered host buffers: ProcessBlocks reads and writes them in place
    T = 300
    bfar, bnear = synth_streams([4100, 4101, 4102], T, fs)
    bout = np.zeros_like(bnear)
    pf, pn, po = (aecm.register_host_buffer(a) for a in (bfar, bnear, bout))
    b = aecm.AecmBatch(3, fs, 1, 2)
    b.process_device(pf, pn, po, T * 64, 64, T)
    b.synchronize()
    for k in range(3):
        assert np.array_equal(bout[k], pyoracle.RefCoreStream(fs, 1, 2).process(bfar[k], bnear[k])), k
    b.close()
    for a in (bfar, bnear, bout):
        aecm.unregister_host_buffer(a)
    with pytest.raises(aecm.AecmError):
        aecm.unregister_host_buffer(bout)                      # not registered any more


def test_ticks_refuse_the_safe_variant_without_poisoning():
    """The tick kernel exists for the fast cross-lane primitives only: a session batch switched to the safe variant gets
    AECM_UNSUPPORTED_FUNCTION_ERROR from its ticks and works again after switching back (it used to be poisoned)."""
    lib = aecm.load()
    sb = aecm.AecmSessions(2, 16000)
    rs = np.random.RandomState(5)
    x = rs.randint(-3000, 3000, size=(2, 160 * 12)).astype(np.int16)
    ref = aecm.AecmSessions(2, 16000)
    for i in range(12):
        sl = slice(i * 160, (i + 1) * 160)
        if i == 6:
            assert lib.WebRtcAecmSessions_SetKernelVariant(sb.h, aecm.KERNEL_SAFE) == 0
            assert sb.tick_host(x[:, sl], x[:, sl])[0] == aecm.ffi.AECM_UNSUPPORTED_FUNCTION_ERROR
            assert lib.WebRtcAecmSessions_SetKernelVariant(sb.h, 7) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
            assert lib.WebRtcAecmSessions_SetKernelVariant(sb.h, aecm.KERNEL_FAST) == 0
        rc, out = sb.tick_host(x[:, sl], x[:, sl])
        rc2, out2 = ref.tick_host(x[:, sl], x[:, sl])
        assert rc == rc2 == 0 and np.array_equal(out, out2), i
    sb.close()
    ref.close()


@_needs_ref
def test_async_tick_event_hooks_order_the_callers_streams():
    """TickAsync's event hooks: the far / near rows are produced by the caller on ITS stream (a slow fill kernel first, so
    that the data is certainly not there yet when the tick is enqueued) and consumed from `out` on another stream; the tick
    must wait for the producer's event and the consumer for the tick's -- no host synchronisation in between.  Every
    session's output of every tick equals a reference session's."""
    import torch
    S, fs, frame, n_calls = 4, 16000, 160, 60
    pairs = [synth_pair(4300 + k, n_calls * frame // 64 + 1, fs, "steady") for k in range(S)]
    far = np.stack([p[0][:n_calls * frame] for p in pairs])
    near = np.stack([p[1][:n_calls * frame] for p in pairs])
    exp = np.stack([pyoracle.RefSession(fs, 1, 3).run(far[k], near[k], frame, 40) for k in range(S)])
    src_far, src_near = torch.from_numpy(far).cuda(), torch.from_numpy(near).cuda()
    dfar, dnear = torch.zeros((S, frame), dtype=torch.int16, device="cuda"), torch.zeros((S, frame), dtype=torch.int16, device="cuda")
    dout = torch.zeros((S, frame), dtype=torch.int16, device="cuda")
    collected = torch.zeros_like(src_near)
    ballast = torch.zeros(1 << 26, dtype=torch.float32, device="cuda")            # ~10 ms of fill work ahead of every row copy
    producer, consumer = torch.cuda.Stream(), torch.cuda.Stream()
    sb = aecm.AecmSessions(S, fs, 1, 3)
    lib = aecm.load()
    torch.cuda.synchronize()
    consumed = None
    for i in range(n_calls):
        sl = slice(i * frame, (i + 1) * frame)
        ready, done = torch.cuda.Event(), torch.cuda.Event()
        done.record(consumer)                           # torch creates the hipEvent_t lazily, at the first record: the tick re-records it
        assert done.cuda_event
        with torch.cuda.stream(producer):
            if consumed is not None:
                producer.wait_event(consumed)              # the rows of the previous tick have been read and its output collected
            ballast.add_(1.0)
            dfar.copy_(src_far[:, sl])
            dnear.copy_(src_near[:, sl])
            ready.record(producer)
        rc = lib.WebRtcAecmSessions_TickAsync(sb.h, dfar.data_ptr(), dnear.data_ptr(), None, dout.data_ptr(), frame, frame, 40, None, None,
                                              None, ready.cuda_event, done.cuda_event)
        assert rc == 0, (i, rc)
        with torch.cuda.stream(consumer):
            consumer.wait_event(done)
            collected[:, sl].copy_(dout)
            consumed = torch.cuda.Event()
            consumed.record(consumer)
    assert sb.synchronize() == 0
    torch.cuda.synchronize()
    assert np.array_equal(collected.cpu().numpy(), exp)
    sb.close()


def test_bulk_state_export_import_equals_the_single_stream_forms():
    """WebRtcAecmBatch_ExportStates / ImportStates (one gather / scatter launch for a range of streams) against the per-stream
    ExportState / ImportState: the same blobs, and a batch restored through the bulk form continues bit-exactly -- through
    host memory, through device memory and through a registered host buffer; a range with one bad blob is refused whole."""
    import struct
    import torch
    fs, T1, T2, S = 16000, 300, 200, 70
    seeds = list(range(8100, 8100 + S))
    far, near = synth_streams(seeds, T1 + T2, fs)
    a = aecm.AecmBatch(S, fs, 1, 2)
    a.process_host(far[:, :T1 * 64], near[:, :T1 * 64])
    n = aecm.load().WebRtcAecmBatch_state_size_bytes()
    blobs = a.export_states(3, S - 5)                                     # streams 3 .. S - 3
    assert blobs.shape == (S - 5, n)
    for k in (0, 1, 33, S - 6):
        assert blobs[k].tobytes() == a.export_state(3 + k), k
    # device memory and a registered host buffer give the same bytes
    dev = torch.empty((S - 5, n), dtype=torch.uint8, device="cuda")
    a.export_states_device(3, S - 5, dev.data_ptr())
    assert np.array_equal(dev.cpu().numpy(), blobs)
    pinned = np.zeros((S - 5, n), dtype=np.uint8)
    alias = aecm.ffi.register_host_buffer(pinned)
    try:
        a.export_states_device(3, S - 5, alias)
        assert np.array_equal(pinned, blobs)
    finally:
        aecm.ffi.unregister_host_buffer(pinned)
    exp_out = a.process_host(far[:, T1 * 64:], near[:, T1 * 64:])       # how the exported streams continue
    # restore into another batch at other slots: host form, then device form
    for form in ("host", "device"):
        b = aecm.AecmBatch(S + 9, fs, 0, 4)
        if form == "host":
            b.import_states(11, blobs)
        else:
            b.import_states_device(11, S - 5, dev.data_ptr())
        fb = np.zeros((S + 9, T2 * 64), np.int16)
        nb = np.zeros_like(fb)
        fb[11:11 + S - 5], nb[11:11 + S - 5] = far[3:S - 2, T1 * 64:], near[3:S - 2, T1 * 64:]
        out_b = b.process_host(fb, nb)
        assert np.array_equal(out_b[11:11 + S - 5], exp_out[3:S - 2]), form
        for k in (0, 20, S - 6):
            assert np.array_equal(b.digest(11 + k), a.digest(3 + k)), (form, k)
        b.close()
    # all or nothing: one bad blob in the middle (history position out of range) -> nothing is touched
    c = aecm.AecmBatch(S, fs)
    before = [c.digest(s) for s in (0, 10, 40, S - 6)]
    bad = blobs.copy()
    struct.pack_into("<i", bad[40], 32 + 12 * 64 * 4 + 4 * 3, 1000)       # S_HISTPOS of blob 40
    lib = aecm.load()
    assert lib.WebRtcAecmBatch_ImportStates(c.h, 0, S - 5, bad.ctypes.data, bad.nbytes) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    dbad = torch.from_numpy(bad).cuda()
    assert lib.WebRtcAecmBatch_ImportStatesDevice(c.h, 0, S - 5, dbad.data_ptr(), bad.nbytes) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    assert all(np.array_equal(c.digest(s), d) for s, d in zip((0, 10, 40, S - 6), before))
    # argument checks: range, size
    assert lib.WebRtcAecmBatch_ExportStates(c.h, S - 2, 3, bad.ctypes.data, 3 * n) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    assert lib.WebRtcAecmBatch_ExportStates(c.h, 0, 3, bad.ctypes.data, 3 * n - 1) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    assert lib.WebRtcAecmBatch_ImportStates(c.h, 0, 1, None, n) == aecm.ffi.AECM_NULL_POINTER_ERROR
    c.close()
    a.close()


def test_bulk_state_export_of_65536_streams_is_fast():
    """The point of the range forms: 65 536 streams are exported with one gather launch -- to device memory, and
    over the link into a registered host buffer -- in well under 50 ms; the per-stream form needs 196 608 blocking copies."""
    import time
    import torch
    S = 65536
    n = aecm.load().WebRtcAecmBatch_state_size_bytes()
    b = aecm.AecmBatch(S, 16000)
    dev = torch.empty((S, n), dtype=torch.uint8, device="cuda")
    b.export_states_device(0, S, dev.data_ptr())                          # warm-up (first touch of 1.06 GB)
    t0 = time.perf_counter()
    b.export_states_device(0, S, dev.data_ptr())
    t_dev = time.perf_counter() - t0
    assert dev[S - 1].cpu().numpy().tobytes() == b.export_state(S - 1)
    assert t_dev < 0.050, t_dev
    host = np.zeros((S, n), dtype=np.uint8)
    alias = aecm.ffi.register_host_buffer(host)
    try:
        b.export_states_device(0, S, alias)
        t0 = time.perf_counter()
        b.export_states_device(0, S, alias)
        t_link = time.perf_counter() - t0
        assert host[S // 2].tobytes() == b.export_state(S // 2)
    finally:
        aecm.ffi.unregister_host_buffer(host)
    print(f"\n65 536 states: {t_dev * 1e3:.2f} ms to device memory, {t_link * 1e3:.2f} ms into a registered host buffer "
          f"({S * n / t_link / 1e9:.1f} GB/s)")
    assert t_link < 0.050, t_link
    b.close()


@_needs_ref
@pytest.mark.parametrize("fs,frame,with_clean", [(16000, 160, 0), (8000, 80, 1), (16000, 80, 0)])
def test_session_migrates_mid_call_between_session_batches(fs, frame, with_clean):
    """WebRtcAecmSessions_ExportSession / ImportSession: live sessions are moved mid-call -- right after a far-end underrun,
    in the middle of the delay compensation that follows start-up, during a run of underruns, and late in steady state -- into
    another AecmSessions object of another size and another age (its near-end ring position differs), into whatever slot.
    From then on the moved session, the session it was copied from (which keeps running) and the reference session that
    received exactly the same calls must agree sample for sample and code for code."""
    S, S2 = 5, 3
    n_calls = 5 * fs // frame // 2                                        # 2.5 s
    pairs = [synth_pair(1300 + k, n_calls * frame // 64 + 1, fs, "mixed") for k in range(S)]
    far = np.stack([p[0][:n_calls * frame] for p in pairs])
    near = np.stack([p[1][:n_calls * frame] for p in pairs])
    clean = synth_clean(near) if with_clean else None
    pats = [call_pattern(90 + k, n_calls) for k in range(S)]
    refs = [pyoracle.RefSession(fs, 1, 3) for _ in range(S)]
    sa = aecm.AecmSessions(S, fs, 1, 3)
    sb = aecm.AecmSessions(S2, fs, 0, 1)
    rs = np.random.RandomState(3)
    # the other object has a life of its own (other age: its near-end ring position is not sa's)
    for _ in range(7):
        z = rs.randint(-3000, 3000, size=(S2, frame)).astype(np.int16)
        sb.tick_host(z, z, 40, z if with_clean else None)
    start_calls = 7 if (fs, frame) == (16000, 160) else 4                 # the first Process after start-up with ms = 40 (+ jitter)
    underrun = int(np.nonzero(pats[2][1] == 0)[0][0])
    run_start = int(np.nonzero((np.arange(n_calls) % 211 >= 205) & (np.arange(n_calls) > 100))[0][0])
    moves = {start_calls + 1: (0, 1), underrun + 1: (2, 0), run_start + 3: (3, 2), n_calls - 40: (4, 1)}    # tick -> (session of sa, slot of sb)
    where = {}                                                            # slot of sb -> session of sa it now mirrors
    n_size = aecm.load().WebRtcAecmSessions_session_size_bytes()
    for i in range(n_calls):
        sl = slice(i * frame, (i + 1) * frame)
        if i in moves:
            k, slot = moves[i]
            rc, snap = sa.export_session(k)
            assert rc == 0 and len(snap) == n_size
            assert sb.import_session(slot, snap) == 0
            rc2, snap2 = sb.export_session(slot)                         # a snapshot does not depend on the object it is in
            assert rc2 == 0 and snap2 == snap, (i, k, slot)
            where[slot] = k
        ms = np.array([pats[k][0][i] for k in range(S)], dtype=np.int16)
        fl = np.array([0 if pats[k][1][i] else aecm.ffi.SESSION_NO_FAREND for k in range(S)], dtype=np.uint8)
        rc, out, codes = sa.tick_host_per_session(far[:, sl], near[:, sl], ms, None if clean is None else clean[:, sl], flags=fl)
        # sb: moved sessions get the calls of the session they mirror; the other slots idle on noise
        fb = rs.randint(-3000, 3000, size=(S2, frame)).astype(np.int16)
        nb = fb.copy()
        cb = fb.copy() if with_clean else None
        msb = np.full(S2, 40, dtype=np.int16)
        flb = np.zeros(S2, dtype=np.uint8)
        for slot, k in where.items():
            fb[slot], nb[slot], msb[slot], flb[slot] = far[k, sl], near[k, sl], ms[k], fl[k]
            if with_clean:
                cb[slot] = clean[k, sl]
        rcb, outb, codesb = sb.tick_host_per_session(fb, nb, msb, cb, flags=flb)
        for k in range(S):
            if not fl[k]:
                assert refs[k].buffer_farend(far[k, sl]) == 0
            rc1, o1 = refs[k].process(near[k, sl], None if clean is None else clean[k, sl], int(ms[k]))
            assert codes[k] == rc1 and np.array_equal(out[k], o1), (fs, frame, i, k)
        for slot, k in where.items():
            assert codesb[slot] == codes[k] and np.array_equal(outb[slot], out[k]), (fs, frame, i, slot, k)
    assert len(where) == 3 and set(where.values()) == {2, 3, 4}        # slot 1 was overwritten by the last move
    # refusals leave the slot as it was: wrong size, bad magic, another rate, an impossible wrapper state
    import struct
    rc, snap = sa.export_session(1)
    lib = aecm.load()
    assert lib.WebRtcAecmSessions_ImportSession(sb.h, 0, snap, len(snap) - 1) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    bad = bytearray(snap); bad[0] ^= 0xff
    assert sb.import_session(0, bytes(bad)) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    bad = bytearray(snap); struct.pack_into("<I", bad, 8, 24000 - fs)
    assert sb.import_session(0, bytes(bad)) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    flow0 = 32 + aecm.load().WebRtcAecmBatch_state_size_bytes()
    bad = bytearray(snap); struct.pack_into("<i", bad, flow0 + 4 * 15, struct.unpack_from("<i", bad, flow0 + 4 * 16)[0] + 6400)   # F_FRM_POS 100 blocks ahead of F_BLK_POS
    assert sb.import_session(0, bytes(bad)) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    bad = bytearray(snap); struct.pack_into("<i", bad, flow0 + 4 * 9, 2)                                                            # F_EC_STARTUP is a flag
    assert sb.import_session(0, bytes(bad)) == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    assert sb.import_session(S2, snap) == aecm.ffi.AECM_BAD_PARAMETER_ERROR and sb.export_session(-1)[0] == aecm.ffi.AECM_BAD_PARAMETER_ERROR
    rc, again = sb.export_session(0)
    assert rc == 0 and again == sa.export_session(2)[1]                  # slot 0 still mirrors session 2, tick for tick
    sa.close()
    sb.close()
