"""The C-ABI shared library loads and exports every symbol include/*.h declares (no compute calls:
those need a GPU and live in the `-m gpu` tests)."""
import ctypes
import re
from pathlib import Path

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


def test_product_does_not_use_the_oracle():
    """The shipped path must never import, include, link or load anything under oracle/."""
    banned = ("aecm_oracle.h", "aecm_oracle_tables.h", "pyoracle", "libaecm_oracle", "libaecm_ref", "import oracle",
              "from oracle", "ref_shim")
    files = [p for p in (ROOT / "webrtc_aecm_amd").rglob("*") if p.is_file() and p.suffix in {".py", ".h", ".cpp", ".hip"}]
    files += list((ROOT / "include").glob("*.h"))
    assert files
    for p in files:
        txt = p.read_text()
        for b in banned:
            assert b not in txt, f"{p} mentions {b}"
