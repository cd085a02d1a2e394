#!/usr/bin/env python3
"""Parity triage: the first block at which the HIP block path and the checker disagree, and in which state field.

The reference's own debugging aid is a set of per-block file dumps behind AEC_DEBUG (echo_control_mobile.cc:105-115).
Here every block is one launch on both sides (state stays resident between launches), and after every block the
24-word state digest (include/aecm_batch.h: WebRtcAecmBatch_GetDigest; oracle/pyoracle.py: DIGEST_NAMES) and the
64 output samples are compared.  Prints the first diverging block, the digest words that differ and the first
differing output sample -- "H(echoFilt) at block 812" tells which phase of aecm_wave.h to read.

    python tools/first_divergence.py [--seed 7] [--blocks 2000] [--fs 16000] [--cng 1] [--echo-mode 3] [--profile mixed]
                                     [--checker reference|oracle] [--engine hip|sim] [--clean] [--far f.raw --near n.raw]

--engine sim runs the kernel's source on the CPU lane simulator (tests/sim) instead of the GPU: same code, no device --
for triaging a kernel change on a box without a GPU.  --perturb-block N flips one input bit on the engine's side only
(a self-check of the tool: it must then report block N).  Exit status 0 = no divergence, 1 = diverged."""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--blocks", type=int, default=2000)
    ap.add_argument("--fs", type=int, default=16000, choices=(8000, 16000))
    ap.add_argument("--cng", type=int, default=1)
    ap.add_argument("--echo-mode", type=int, default=3)
    ap.add_argument("--profile", default=None, help="webrtc_aecm_amd.synth profile (mixed, steady, loud, sparse, silent)")
    ap.add_argument("--far", help="raw int16 little-endian far-end samples instead of the synthetic pair")
    ap.add_argument("--near", help="raw int16 little-endian near-end samples")
    ap.add_argument("--clean", action="store_true", help="also feed a clean near-end input (3/4 of the near end)")
    ap.add_argument("--checker", choices=("reference", "oracle"), default=None, help="default: the reference when oracle/_ref is built")
    ap.add_argument("--engine", choices=("hip", "sim"), default="hip")
    ap.add_argument("--variant", choices=("fast", "safe"), default="fast")
    ap.add_argument("--perturb-block", type=int, default=-1)
    a = ap.parse_args()

    from oracle import pyoracle
    from webrtc_aecm_amd.synth import synth_clean, synth_pair
    if a.far and a.near:
        far = np.fromfile(a.far, dtype="<i2")
        near = np.fromfile(a.near, dtype="<i2")
        n = min(far.size, near.size) // 64
        far, near = far[:n * 64].copy(), near[:n * 64].copy()
    else:
        far, near = synth_pair(a.seed, a.blocks, a.fs, a.profile)
        n = a.blocks
    clean = synth_clean(near) if a.clean else None
    use_ref = (a.checker == "reference") or (a.checker is None and pyoracle.have_reference())
    chk = pyoracle.RefCoreStream(a.fs, a.cng, a.echo_mode) if use_ref else pyoracle.OracleStream(a.fs, a.cng, a.echo_mode)

    if a.engine == "hip":
        import webrtc_aecm_amd as aecm
        eng = aecm.AecmBatch(1, a.fs, a.cng, a.echo_mode, variant=aecm.KERNEL_FAST if a.variant == "fast" else aecm.KERNEL_SAFE)

        def run(f, d, c):
            return eng.process_host(f[None, :], d[None, :], None if c is None else c[None, :])[0]

        def digest():
            return eng.digest(0)
    else:
        import simlib
        eng = simlib.SimStream(a.fs, a.cng, a.echo_mode)

        def run(f, d, c):
            return eng.process(f, d, c)

        def digest():
            return eng.digest()

    print(f"{n} blocks, fs {a.fs}, cng {a.cng}, echoMode {a.echo_mode}; engine: {a.engine}"
          f"{' (' + a.variant + ')' if a.engine == 'hip' else ''}; checker: {'reference (oracle/_ref)' if use_ref else 'oracle restatement'}")
    for b in range(n):
        sl = slice(b * 64, (b + 1) * 64)
        f, d = far[sl].copy(), near[sl].copy()
        c = None if clean is None else clean[sl].copy()
        exp = chk.process(f, d) if c is None else chk.process_block_clean(f, d, c)
        if b == a.perturb_block:
            d = d.copy()
            d[17] ^= 0x10
        got = run(f, d, c)
        dg, de = digest(), chk.digest()
        if not np.array_equal(got, exp) or not np.array_equal(dg, de):
            words = [f"[{i}] {pyoracle.DIGEST_NAMES[i]}: engine {int(dg[i]):#010x} != checker {int(de[i]):#010x}"
                     for i in np.nonzero(dg != de)[0]]
            print(f"FIRST DIVERGENCE at block {b} (sample {b * 64}, {b * 64 / a.fs:.3f} s)")
            bad = np.nonzero(got != exp)[0]
            if bad.size:
                print(f"  output: {bad.size} of 64 samples differ, first at index {int(bad[0])}: engine {int(got[bad[0]])} != checker {int(exp[bad[0]])}")
            else:
                print("  output: identical (the state diverged before the output did)")
            print("  state digest words that differ:" if words else "  state digest: identical")
            for w in words:
                print("    " + w)
            return 1
    print(f"no divergence in {n} blocks (outputs and all 24 digest words equal after every block)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
