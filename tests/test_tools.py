"""tools/first_divergence.py, the parity triage tool (the reference's AEC_DEBUG dump analogue), on the CPU lane simulator:
it must report nothing on an intact build and the exact block when one side's input is perturbed."""
import subprocess
import sys
from pathlib import Path

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
