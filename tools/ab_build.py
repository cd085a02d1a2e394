#!/usr/bin/env python3
"""Build A/B variants of the HIP library next to the shipping one (webrtc_aecm_amd/_lib/ab_<name>.so,
git-ignored, travels with gpurun) for kernel experiments:

    python tools/ab_build.py name1="-DFOO=1" name2="-DFOO=2 -mllvm -bar"
    AECM_LIB_PATH=webrtc_aecm_amd/_lib/ab_name1.so python bench.py --no-cpu-baseline
"""
import subprocess, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from webrtc_aecm_amd import build as B  # noqa: E402


def main():
    procs = []
    for arg in sys.argv[1:]:
        name, _, flags = arg.partition("=")
        out = B.LIB_DIR / f"ab_{name}.so"
        cmd = [B._hipcc(), *B.HIPCC_FLAGS, *flags.split(), *[str(B.CSRC / s) for s in B.SOURCES], "-o", str(out)]
        procs.append((name, out, subprocess.Popen(cmd, cwd=str(B.CSRC))))
    bad = 0
    for name, out, p in procs:
        rc = p.wait()
        print(f"{name}: {'ok ' + str(out) if rc == 0 else 'FAILED'}")
        bad += rc != 0
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
