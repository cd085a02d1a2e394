"""Host-side code under sanitizers (ASan / UBSan on the host code).

The pointer arithmetic of the product's host layers -- session wrapper and frame adapter (aecm_session.cpp,
aecm_session_flow.h), the device plan's position arithmetic (aecm_flow_plan.h), the recordings schedule
(aecm_schedule.cpp), the host state image / constants blob (aecm_host_state.cpp) -- and the block DSP source itself
(aecm_wave.h on the CPU lane simulator) are compiled with -fsanitize=address,undefined -fno-sanitize-recover=all into
tests/_build/libaecm_sim_san.so, and the tests that drive them run on that library in a child process with the sanitizer
runtimes preloaded.  Any report aborts the child.  (The GPU-side counterpart: tests/test_gpu_parity.py::test_c_abi_under_ubsan.)"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

# the host layers in full, and the block DSP source through the session tests (which run whole recordings through it)
SELECTED = "session or schedule or sample_tags or constants or echo_path or reciprocal"


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_host_layers_and_block_dsp_under_asan_ubsan(tmp_path):
    import pytest
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer runtimes not installed")
    env = dict(os.environ, AECM_SIM_SANITIZE="1", LD_PRELOAD=f"{asan}:{ubsan}",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",      # the interpreter's own arenas are not ours to audit
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", PYTHONMALLOC="malloc")
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_sim.py"), "-x", "-q", "-p", "no:cacheprovider",
                        "-k", SELECTED], env=env, capture_output=True, text=True, timeout=3000, cwd=str(ROOT))
    log = r.stdout + r.stderr
    (tmp_path / "sanitizer.log").write_text(log)
    if os.environ.get("AECM_SANITIZER_LOG"):                     # keep the evidence (profiles/r03_sanitizers_cpu.log)
        Path(os.environ["AECM_SANITIZER_LOG"]).write_text(log)
    assert r.returncode == 0, log[-4000:]
    assert "runtime error" not in log and "AddressSanitizer" not in log, log[-4000:]
    assert " passed" in log and "libaecm_sim_san.so" in "".join(p.name for p in (ROOT / "tests" / "_build").iterdir())
