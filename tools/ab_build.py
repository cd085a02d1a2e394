#!/usr/bin/env python3
"""Build A/B variants of the HIP library next to the shipping one (webrtc_aecm_amd/_lib/ab_<name>.so,
git-ignored, travels with gpurun) for kernel experiments:

    python tools/ab_build.py name1="-DFOO=1" name2="-DFOO=2 -mllvm -bar"
    AECM_LIB_PATH=webrtc_aecm_amd/_lib/ab_name1.so python bench.py --no-cpu-baseline

The extra flags go to every source, after the per-source flags of build.SOURCE_FLAGS; "name:source.hip=flags" gives them
to one source only.
"""
import subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from webrtc_aecm_amd import build as B  # noqa: E402


def build_variant(arg):
    name, _, flags = arg.partition("=")
    name, _, only = name.partition(":")
    out = B.LIB_DIR / f"ab_{name}.so"
    compile_flags = [f for f in B.HIPCC_FLAGS if f != "-shared"]
    with tempfile.TemporaryDirectory() as td:
        objs = []

        def one(src):
            obj = Path(td) / (src + ".o")
            extra = flags.split() if (not only or only == src) else []
            rc = subprocess.call([B._hipcc(), *compile_flags, *B.SOURCE_FLAGS.get(src, []), *extra, "-c", str(B.CSRC / src), "-o", str(obj)],
                                 cwd=str(B.CSRC))
            return obj, rc
        with ThreadPoolExecutor(max_workers=4) as ex:
            res = list(ex.map(one, B.SOURCES))
        if any(rc for _, rc in res):
            return name, out, 1
        objs = [str(o) for o, _ in res]
        rc = subprocess.call([B._hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", str(out)], cwd=str(B.CSRC))
    return name, out, rc


def main():
    with ThreadPoolExecutor(max_workers=3) as ex:
        results = list(ex.map(build_variant, sys.argv[1:]))
    bad = 0
    for name, out, rc in results:
        print(f"{name}: {'ok ' + str(out) if rc == 0 else 'FAILED'}")
        bad += rc != 0
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
