"""The C-ABI shared library loads and exports every symbol include/*.h declares (no compute calls:
those need a GPU and live in the `-m gpu` tests)."""
import ctypes
import re
from pathlib import Path

import ctypes as C

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared(header):
    txt = (ROOT / "include" / header).read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(WebRtcAecm(?:Batch|Sessions)?_\w+)\s*\(", txt)))


def test_library_builds_and_exports_all_declared_symbols():
    from webrtc_aecm_amd import build, ffi
    lib_path = build.build()
    assert lib_path.exists()
    lib = ctypes.CDLL(str(lib_path))
    session = _declared("echo_control_mobile.h")
    batch = _declared("aecm_batch.h")
    assert sorted(session) == sorted(ffi.SESSION_SYMBOLS)
    assert sorted(batch) == sorted(ffi.BATCH_SYMBOLS + ffi.SESSIONS_SYMBOLS)
    for name in session + batch:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"


def test_reference_abi_surface_is_complete():
    # the 10 functions + config struct of the reference header (aecm/echo_control_mobile.h:46-202)
    names = _declared("echo_control_mobile.h")
    assert names == sorted(["WebRtcAecm_Create", "WebRtcAecm_Free", "WebRtcAecm_Init", "WebRtcAecm_BufferFarend",
                            "WebRtcAecm_GetBufferFarendError", "WebRtcAecm_Process", "WebRtcAecm_set_config",
                            "WebRtcAecm_InitEchoPath", "WebRtcAecm_GetEchoPath", "WebRtcAecm_echo_path_size_bytes"])


def test_no_cpu_fallback_without_gpu(gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    import webrtc_aecm_amd as aecm
    with pytest.raises(RuntimeError):
        aecm.Aecm()
    with pytest.raises(RuntimeError):
        aecm.AecmBatch(4)
    lib = aecm.load()
    assert lib.WebRtcAecm_echo_path_size_bytes() == 130
    assert lib.WebRtcAecm_Init(None, 16000) == -1


def test_launch_form_rules_without_a_device():
    """Which kernel a launch takes is host logic (WebRtcAecmBatch_DescribeLaunchFor / DescribeLaunchDetail: the engine's own rules for
    a device of that many CUs; tests/test_gpu_parity.py::test_launch_form_by_size asks a live engine the same questions): one stream ->
    one wavefront; up to 4 x 4 x CUs streams and at least three blocks -> pipelined, the shape by streams per CU (sixteen waves per
    workgroup up to eight streams per CU -- two such workgroups --, eight waves up to twelve, six -- balanced in launches of >= 128
    blocks -- above); more -> the chunk queue where the launch is at least two chunks long; never pipelined with a clean input."""
    import webrtc_aecm_amd as aecm
    for cus in (256, 304, 64):
        pipe_max, resident, rotation, tail_max = cus * 16, cus * 28, cus * 24, cus * 12
        for S, T, clean, want in ((1, 300, False, (0, 0)), (2, 300, False, (3, 0x1a02)), (cus * 4, 300, False, (3, 0x1a02)), (cus * 4, 2, False, (0, 0)),
                                  (cus * 4, 3, False, (3, 0x1a02)), (cus * 4 + 1, 300, False, (3, 0x1a02)), (cus * 8, 300, False, (3, 0x1a02)),
                                  (cus * 8 + 1, 300, False, (3, 0x402)), (tail_max, 300, False, (3, 0x402)), (tail_max + 1, 300, False, (3, 0x500)),
                                  (pipe_max, 100, False, (3, 0)),
                                  (pipe_max, 300, False, (3, 0x500 if pipe_max <= 5120 else 0)),      # (the balance's monitor reads at most 1 280 workgroups' words)
                                  (pipe_max, 300, True, (0, 0)),
                                  (pipe_max + 1, 300, False, (2, 32)), (pipe_max + 1, 63, False, (0, 0)), (rotation + 1, 63, False, (1, 0)),
                                  (resident, 64, True, (2, 32)), (resident + 1, 255, False, (1, 0)), (resident + 1, 256, False, (2, 128))):
            assert aecm.describe_launch_for(S, cus, T, clean) == want, (cus, S, T, clean, aecm.describe_launch_for(S, cus, T, clean))
            d = aecm.describe_launch_detail(S, T, cus, clean)
            assert (d["form"], d["chunk_blocks"] if d["form"] == 2 else d["shape"] if d["form"] == 3 else 0) == want, (cus, S, T, d)
    with pytest.raises(aecm.AecmError):
        aecm.describe_launch_for(0, 256, 300)


def test_launch_policy_is_one_value_and_the_library_reads_no_environment(monkeypatch):
    """The launch policy through the C ABI (AecmLaunchPolicy): the default derives from the CU count alone, wishes set on it reach
    the launch rules, invalid values are refused; and the shipped build consults no environment variable -- neither for the policy
    (the AECM_* wishes of an -DAECM_EXPERIMENTS build are ignored) nor anywhere else (no getenv outside AECM_EXPERIMENTS)."""
    import re
    import webrtc_aecm_amd as aecm
    from webrtc_aecm_amd import ffi
    for k, v in (("AECM_PIPE_GAIN", "0"), ("AECM_PIPE_DELAY", "0"), ("AECM_PIPELINED", "0"), ("AECM_QUEUE_CHUNK", "7"), ("AECM_PIPE_SPREAD", "0")):
        monkeypatch.setenv(k, v)
    p = aecm.default_launch_policy(256)
    assert p.as_dict() == dict(struct_size=C.sizeof(ffi.AecmLaunchPolicy), compute_units=256, queue_chunk_blocks=128, queue_chunk_explicit=0,
                               queue_min_streams=-1, pipelined_min_streams=2, pipelined_min_blocks=3, pipelined_max_streams=4096,
                               resident_waves=7168, rotation_stream_limit=6144, pipe_tail_waves=-1, pipe_front_waves=-1, pipe_raw=-1,
                               pipe_delay_waves=-1, pipe_gain_waves=-1, pipe_spread=1, pipe_wgs_per_cu=0, pipe_rot=-1, pipe_prio=-1)
    assert aecm.describe_launch_for(1024, 256, 300) == (3, 0x1a02)                      # the environment above changed nothing
    # wishes on a policy
    shape = lambda **kw: (lambda q: [setattr(q, k, v) for k, v in kw.items()] and aecm.describe_launch_detail(1024, 300, policy=q))(aecm.default_launch_policy(256))
    assert shape(pipe_gain_waves=0)["shape"] == 0x802
    assert shape(pipe_delay_waves=0)["shape"] == 0x2 and shape(pipe_delay_waves=0, pipe_raw=1, pipe_front_waves=4)["shape"] == 0x602
    assert shape(pipe_tail_waves=0)["shape"] == 0x0
    # wishes that name a kernel the library does not carry land on the nearest one it does (never on a launch error):
    # four front waves without the raw hand-over -> with it, for every size and CU count
    for cus, S in ((256, 1024), (256, 2048), (64, 300), (304, 2000)):
        q = aecm.default_launch_policy(cus)
        q.pipe_raw, q.pipe_delay_waves, q.pipe_front_waves = 0, 0, 4
        assert aecm.describe_launch_detail(S, 300, policy=q)["shape"] == 0x602, (cus, S)
    built = {0x0, 0x500, 0x2, 0x402, 0x602, 0x802, 0x1a02}
    for tail in (-1, 0, 2):
        for front in (-1, 2, 4):
            for raw in (-1, 0, 1):
                for delay in (-1, 0, 2, 4):
                    for gain in (-1, 0, 4):
                        for S in (7, 1024, 1500, 2048, 2500, 3072, 3500, 4096):
                            q = aecm.default_launch_policy(256)
                            q.pipe_tail_waves, q.pipe_front_waves, q.pipe_raw, q.pipe_delay_waves, q.pipe_gain_waves = tail, front, raw, delay, gain
                            d = aecm.describe_launch_detail(S, 300, policy=q)
                            assert d["form"] == 3 and d["shape"] in built, (tail, front, raw, delay, gain, S, hex(d["shape"]))
    assert shape(pipelined_min_streams=0)["form"] == 0
    assert shape(queue_min_streams=0, pipelined_min_streams=5000) == dict(form=2, chunk_blocks=32, shape=0, workgroups=256, waves_per_workgroup=4,
                                                                           workgroups_per_cu=7, rounds_x1000=142, cu_load_evenness_x1000=1000)
    # every CU its full count of workgroups: 1 536 streams = two sixteen-wave workgroups of three streams per CU; without the spread, 384 of four
    d = aecm.describe_launch_detail(1536, 300, 256)
    assert (d["workgroups"], d["waves_per_workgroup"], d["workgroups_per_cu"], d["rounds_x1000"]) == (512, 16, 2, 1000)
    assert shape(pipe_spread=0)["workgroups"] == 256 and aecm.describe_launch_detail(2560, 300, 256)["workgroups"] == 768
    # a tick of 65 536 sessions: 16 384 workgroups on 1 792 places (the last round 14 % full)
    assert aecm.describe_tick(65536, 256) == dict(form=0, chunk_blocks=0, shape=0, workgroups=16384, waves_per_workgroup=4, workgroups_per_cu=7,
                                                  rounds_x1000=9142, cu_load_evenness_x1000=1000)
    # a pipelined launch keeps a stream on one CU: batches that are multiples of the CU count load every CU alike, one stream more does not
    assert [aecm.describe_launch_detail(S, 300, 256)["cu_load_evenness_x1000"] for S in (256, 1024, 1025, 1280, 2049, 3072, 4095, 100)] == \
        [1000, 1000, 800, 1000, 889, 1000, 999, 1000]
    # refused: another struct size, values outside what the kernels exist for
    for bad in (dict(struct_size=8), dict(pipe_front_waves=3), dict(pipe_gain_waves=2), dict(queue_chunk_blocks=-1), dict(pipelined_max_streams=5000),
                dict(pipe_rot=4096), dict(pipe_prio=256)):
        q = aecm.default_launch_policy(256)
        for k, v in bad.items():
            setattr(q, k, v)
        with pytest.raises(aecm.AecmError):
            aecm.describe_launch_detail(1024, 300, policy=q)
    # no getenv in the default build: every use in the sources sits between #if defined(AECM_EXPERIMENTS) / AECM_PIPE_TRACE and its #endif
    for src in sorted((ROOT / "webrtc_aecm_amd" / "csrc").glob("*")):
        depth_exp, stack = 0, []
        for n, line in enumerate(src.read_text().splitlines(), 1):
            t = line.strip()
            if t.startswith("#if"):
                stack.append(bool(re.search(r"AECM_EXPERIMENTS|AECM_PIPE_TRACE", t)))
            elif t.startswith("#endif") and stack:
                stack.pop()
            elif "getenv" in t and not t.startswith("//"):
                assert any(stack), f"{src.name}:{n}: getenv in the default build"


def test_cmake_build_equals_the_python_recipe(tmp_path):
    """The C/C++-native recipe (top-level CMakeLists.txt: `cmake -S . -B build && cmake --build build`, no Python) and
    webrtc_aecm_amd/build.py must produce the same product: every kernel of both libraries with the same instruction stream
    (isa_census fingerprints -- the per-source -mllvm flags are worth 9 % of the frame rate), the same exported C ABI, and the
    command-line tool."""
    import shutil
    import subprocess
    if not shutil.which("cmake"):
        pytest.skip("no cmake on this machine")
    from webrtc_aecm_amd import build, ffi, isa_census
    build.build()
    out = tmp_path / "build"
    subprocess.run(["cmake", "-S", str(ROOT), "-B", str(out)], check=True, capture_output=True)
    subprocess.run(["cmake", "--build", str(out), "-j", "8"], check=True, capture_output=True)
    for ours, theirs, strict in ((build.LIB, out / "libaecm_mi355x.so", True), (build.LIB_CHECKED, out / "libaecm_mi355x_checked.so", False)):
        a = isa_census.census_of_text(isa_census.disassemble(ours))
        b = isa_census.census_of_text(isa_census.disassemble(theirs))
        assert len(a) >= 25 and set(a) == set(b)
        if strict:      # the product: instruction for instruction, operand for operand
            assert {k: v["fingerprint"] for k, v in a.items()} == {k: v["fingerprint"] for k, v in b.items()}, theirs.name
        # the audit twin: the same instructions, opcode by opcode (this compiler allocates the registers of its largest kernel --
        # the self test with every check compiled in -- differently from one run to the next, same command line: 2 of 7 builds)
        assert {k: v["opcodes"] for k, v in a.items()} == {k: v["opcodes"] for k, v in b.items()}, theirs.name
    lib = ctypes.CDLL(str(out / "libaecm_mi355x.so"))
    for name in ffi.SESSION_SYMBOLS + ffi.BATCH_SYMBOLS + ffi.SESSIONS_SYMBOLS:
        assert hasattr(lib, name), name
    assert (out / "aecm_run").exists()
    r = subprocess.run([str(out / "aecm_run")], capture_output=True, text=True)
    assert "usage : aecm_run" in r.stdout


def test_product_does_not_use_the_oracle():
    """The shipped path must never import, include, link or load anything under oracle/."""
    banned = ("aecm_oracle.h", "aecm_oracle_tables.h", "pyoracle", "libaecm_oracle", "libaecm_ref", "import oracle",
              "from oracle", "ref_shim", "ref_wavdec")
    files = [p for p in (ROOT / "webrtc_aecm_amd").rglob("*") if p.is_file() and p.suffix in {".py", ".h", ".cpp", ".hip"}]
    files += list((ROOT / "include").glob("*.h"))
    assert files
    for p in files:
        txt = p.read_text()
        for b in banned:
            assert b not in txt, f"{p} mentions {b}"


def test_block_kernel_register_budget(tmp_path):
    """The headline kernels are built for 7 waves per SIMD (72 VGPRs; the small-launch variants for 6 = 80): the fast
    variants must fit without spilling to scratch, or the occupancy the measurements rely on is silently gone."""
    import re
    import subprocess
    from webrtc_aecm_amd import build
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    text = ""
    for src in build.KERNEL_SOURCES:                      # each unit with the flags the build gives it
        out = tmp_path / (src + ".s")
        subprocess.check_call([build._hipcc(), *flags, *build.SOURCE_FLAGS.get(src, []), "-S", "--cuda-device-only", f"-I{build.CSRC}",
                               str(build.CSRC / src), "-o", str(out)], stderr=subprocess.DEVNULL)
        text += out.read_text()
    for has_clean, phase_prio in (("0", "1"), ("1", "1"), ("0", "0"), ("1", "0")):     # <fast, clean, issue priority by phase>
        m = re.search(r"^_ZN4aecm19aecm_process_kernelILb1ELb%sELb%sEEE\w*:.*\n" % (has_clean, phase_prio), text, re.M)
        assert m, "fast block kernel not found in the device assembly"
        body = text[m.end():]
        body = body[:body.index(".end_amdhsa_kernel")]
        vgprs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        scratch = len(re.findall(r"^\s*scratch_(load|store)", body, re.M))
        # phase-priority variants (launches larger than the chip): 7 waves per SIMD = 72 VGPRs; rotation variants (launches
        # of at most 6 waves per SIMD): 80
        assert vgprs <= (72 if phase_prio == "1" else 80) and scratch == 0, (has_clean, phase_prio, vgprs, scratch)
    # The chunk-queue kernels (launches larger than the chip): the same 7 waves per SIMD, no scratch either (the per-lane
    # address halves are re-formed per item, wave_gfx950.h: stream_lane_id).
    for has_clean, max_scratch in (("0", 0), ("1", 0)):
        m = re.search(r"^_ZN4aecm25aecm_process_queue_kernelILb%sEEE\w*:.*\n" % has_clean, text, re.M)
        assert m, "chunk-queue kernel not found in the device assembly"
        body = text[m.end():]
        body = body[:body.index(".end_amdhsa_kernel")]
        vgprs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        scratch = len(re.findall(r"^\s*scratch_(load|store)", body, re.M))
        assert vgprs <= 72 and scratch <= max_scratch, (has_clean, vgprs, scratch)
        # state and history travel at agent scope: sc1 on every access to them, and no release / acquire fence (a write-back
        # or invalidate of the XCD's whole L2 per item)
        assert len(re.findall(r"global_(load|store)_\w+ .* sc1", body)) >= 30
        assert not re.search(r"buffer_(wbl2|inv)", body), "an agent-scope fence crept into the chunk-queue kernel"
    # The pipelined kernel (launches the chip holds at once; 24 of its waves per CU): the same budget, no scratch; its two
    # roles meet at workgroup barriers only (no polling loops: no s_sleep).
    for tail_waves, balance, raw, front, delay, gain in ((0, 0, 0, 2, 0, 0), (0, 1, 1, 2, 0, 0), (2, 0, 0, 2, 0, 0), (2, 0, 1, 2, 0, 0), (2, 0, 1, 4, 0, 0),
                                                         (2, 0, 0, 2, 4, 0), (2, 0, 0, 4, 2, 4)):       # instantiations the launcher uses (template arguments)
        m = re.search(r"^_ZN4aecm29aecm_process_pipelined_kernelILi%dELb%dELb%dELi%dELi%dELi%dEEE\w*:.*\n" % (tail_waves, balance, raw, front, delay, gain), text, re.M)
        assert m, "pipelined kernel not found in the device assembly"
        body = text[m.end():]
        body = body[:body.index(".end_amdhsa_kernel")]
        vgprs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        assert vgprs <= 72 and not re.search(r"^\s*scratch_(load|store)", body, re.M), (tail_waves, balance, raw, front, delay, gain, vgprs)
        assert len(re.findall(r"^\s*s_barrier", body, re.M)) >= 4 and not re.search(r"^\s*s_sleep", body, re.M)
    # The tick kernel: the engine's 64 scalar state words must arrive through scalar loads.  A conditional fence, or a
    # store wider than the rings' int16 (vector types alias everything), ahead of load_state silently turns them into
    # vector loads + v_readfirstlane (profiles/r02_experiments.md).  The only 16-byte vector loads it may contain are those
    # of the LDS table fill (20 KB by the 256 threads of a 4-session workgroup: six).
    for has_clean in ("0", "1"):
        m = re.search(r"^_ZN4aecm21aecm_tick_flow_kernelILb%sEEE\w*:.*\n" % has_clean, text, re.M)
        assert m, "tick kernel not found in the device assembly"
        body = text[m.end():]
        body = body[:body.index(".end_amdhsa_kernel")]
        wide_loads = len(re.findall(r"^\s*global_load_dwordx4", body, re.M))
        scalar_loads = len(re.findall(r"^\s*s_load_", body, re.M))
        assert wide_loads <= 6 and scalar_loads >= 14, (has_clean, wide_loads, scalar_loads)


def test_kernels_fit_the_residency_the_launch_policy_counts_on(tmp_path):
    """The launch policy counts on a residency per kernel -- 28 waves of the one-wave-per-stream and tick kernels per CU, 2 x 16 of
    the sixteen-wave pipelined kernel, 3 x 8 and 4 x 6 of the eight- and six-wave ones (WebRtcAecmBatch_DescribeLaunchDetail:
    workgroups_per_cu x waves_per_workgroup).  What the hardware admits follows from the code object (MI355X_MICROARCH.md,
    "Residency": waves per SIMD = min(8, 512 / VGPRs rounded up to 8, 800 / (SGPRs rounded up to 16 + 16)) -- the compiler's own
    occupancy line is one too high at 82-96 SGPRs, which is how two sixteen-wave workgroups never shared a CU in round 5): the
    kernel metadata of the built library must admit what the policy assumes."""
    import subprocess
    import webrtc_aecm_amd as aecm
    from webrtc_aecm_amd import build, isa_census
    build.build()
    lib = tmp_path / build.LIB.name
    lib.write_bytes(build.LIB.read_bytes())
    subprocess.run([isa_census._tool("llvm-objdump"), "--offloading", str(lib)], check=True, capture_output=True)
    meta = ""
    for obj in sorted(tmp_path.glob(lib.name + ".*amdgcn*gfx950*")):
        meta += subprocess.run([isa_census._tool("llvm-readelf"), "--notes", str(obj)], check=True, capture_output=True, text=True).stdout
    kernels = {}
    for block in re.split(r"\n\s*- \.agpr_count:", meta)[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        kernels[name] = (int(re.search(r"\.sgpr_count:\s+(\d+)", block).group(1)), int(re.search(r"\.vgpr_count:\s+(\d+)", block).group(1)),
                         int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1)))
    assert len(kernels) >= 25

    def admitted(sym_regex):
        hits = [(k, v) for k, v in kernels.items() if re.search(sym_regex, k)]
        assert len(hits) == 1, (sym_regex, [k for k, _ in hits])
        (sgpr, vgpr, scratch) = hits[0][1]
        return min(8, 512 // (-(-vgpr // 8) * 8), 800 // (-(-sgpr // 16) * 16 + 16)), scratch

    cus = 256
    for S, T in ((1000, 300), (2048, 300), (3000, 300), (4096, 300), (5000, 300), (65536, 1280), (6000, 40)):
        d = aecm.describe_launch_detail(S, T, cus)
        need = -(-d["workgroups_per_cu"] * d["waves_per_workgroup"] // 4)                  # waves per SIMD the policy counts on
        waves, scratch = admitted(isa_census.block_kernel(d["form"], False, d["chunk_blocks"] if d["form"] == 2 else d["shape"])[0])
        assert waves >= need and scratch == 0, (S, T, d, waves, scratch)
    d = aecm.describe_tick(65536, cus)
    waves, _ = admitted(r"aecm_tick_flow_kernelILb0E")
    assert waves >= -(-d["workgroups_per_cu"] * d["waves_per_workgroup"] // 4), (d, waves)


def test_forwarder_header_and_unmodified_reference_caller_links():
    """include/aecm/echo_control_mobile.h makes `#include "aecm/echo_control_mobile.h"` (reference main.cc:18) find the
    drop-in declarations; where the reference tree is present its main.cc, unmodified, compiles against include/
    and links to libaecm_mi355x.so (oracle/Makefile: refmain).  The binary is run in the -m gpu tests."""
    import subprocess
    fwd = ROOT / "include" / "aecm" / "echo_control_mobile.h"
    assert fwd.exists() and '#include "../echo_control_mobile.h"' in fwd.read_text()
    if not Path("/root/reference/main.cc").is_file():
        pytest.skip("reference tree not present (GPU box): the prebuilt oracle/_ref/aecm_run_refmain is used")
    from oracle import pyoracle
    from webrtc_aecm_amd import build
    build.build()
    pyoracle.build()
    assert pyoracle.REFMAIN.exists()
    needed = subprocess.run(["ldd", str(pyoracle.REFMAIN)], capture_output=True, text=True).stdout
    assert "libaecm_mi355x.so" in needed and "not found" not in needed
    assert "libaecm_ref" not in needed


def test_isa_census_of_the_built_library():
    """bench.py's provenance check: the block kernel can be found and fingerprinted inside the built .so, it has no
    MFMA and no scratch, and its static mix is the integer VALU + scalar mix DESIGN.md describes."""
    from webrtc_aecm_amd import build, isa_census
    c = isa_census.census(build.build())
    assert isa_census.HEADLINE_KERNEL in c["kernel"] and len(c["fingerprint"]) == 16
    names = [sub for sub, _ in isa_census.BLOCK_KERNELS.values()] + [isa_census.block_kernel(3, False, d)[0] for d in (0, 2, 0x402, 0x602, 0x500)]
    for sub in names:                                                   # every kernel bench.py may name exists in the library
        assert re.search(sub, isa_census.census(build.LIB, sub)["kernel"])
    assert c["counts"]["VALU"] > 800 and c["counts"]["SALU"] > 300
    assert not any(op.startswith(("v_mfma", "scratch_")) for op in c["opcodes"])
    assert c["opcodes"].get("v_dot2c_i32_i16_e32", 0) + c["opcodes"].get("v_dot2_i32_i16", 0) >= 60
    assert 0 < c["valu_fast_class"] < c["counts"]["VALU"]


def test_sanitizer_twin_is_not_part_of_the_shipped_build(monkeypatch, tmp_path):
    """The UBSan twin of the library is opt-in test infrastructure (build_ubsan): its absence must neither make the
    shipped library stale (every load() would retry a failing build on a toolchain without the sanitizer runtime) nor
    be something build() produces."""
    from webrtc_aecm_amd import build
    if build.is_stale():
        pytest.skip("library not built yet")
    monkeypatch.setattr(build, "LIB_UBSAN", tmp_path / "libaecm_mi355x_ubsan.so")
    assert not build.is_stale() and build.ubsan_is_stale()
    assert build.build() == build.LIB and not (tmp_path / "libaecm_mi355x_ubsan.so").exists()
    # a toolchain without the runtime: a clear error from the opt-in entry point, nothing else touched
    monkeypatch.setattr(build, "_ubsan_runtime_dir", lambda hipcc: None)
    before = build.LIB.stat().st_mtime
    with pytest.raises(RuntimeError, match="UBSan runtime"):
        build.build_ubsan()
    assert build.LIB.stat().st_mtime == before
