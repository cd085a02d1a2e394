#!/usr/bin/env python3
"""Static instruction census of the block kernel by phase.

Compiles csrc/aecm_kernels.hip for gfx950 with -DAECM_MARKERS (comment markers at the phase
boundaries of BlockEngine::process_block, see aecm_wave.h) and counts VALU / SALU / LDS / VMEM
instructions between consecutive markers inside the steady-state block loop of
aecm_process_kernel<true,false>.  The marker build perturbs scheduling slightly (markers are
barriers), so totals differ by a few instructions from the shipping build; PMC counts
(profiles/*_rocprof_summary.json) are the ground truth for totals, this is for attribution.

    python tools/isa_phase_breakdown.py [--kernel SUBSTR] [--dump out.s]
"""
import argparse, collections, re, subprocess, sys, tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "webrtc_aecm_amd" / "csrc"
PHASES = {0: "load+prefetch", 1: "far window+FFT+mag", 2: "near window+FFT+mag", 3: "far history/binary far",
          4: "binary near + delay estimator", 5: "aligned far fetch", 6: "energies/VAD", 7: "NLMS channel update",
          8: "store/restore + supgain + Wiener", 9: "NLP + comfort noise prep", 10: "comfort noise",
          11: "IFFT", 12: "synthesis + store", 13: "loop tail"}


def classify(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_barrier", "s_endpgm", "s_sleep")):
            return "SCTL"
        return "SALU" if not op.startswith(("s_load", "s_buffer_load")) else "SMEM"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="aecm_process_kernelILb1ELb0EE")
    ap.add_argument("--dump")
    ap.add_argument("--extra", default="", help="extra hipcc flags")
    a = ap.parse_args()
    sys.path.insert(0, str(ROOT))
    from webrtc_aecm_amd import build as B
    flags = [f for f in B.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] if hasattr(B, "HIPCC_FLAGS") else \
        ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fwrapv", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "k.s"
        cmd = ["/opt/rocm/bin/hipcc", *flags, *a.extra.split(), "-DAECM_MARKERS", "-S", "--cuda-device-only",
               f"-I{CSRC}", str(CSRC / "aecm_kernels.hip"), "-o", str(out)]
        subprocess.check_call(cmd)
        text = out.read_text()
        if a.dump:
            Path(a.dump).write_text(text)
    # isolate the kernel body
    m = re.search(r"^(_Z\w*%s\w*):.*\n" % re.escape(a.kernel), text, re.M)
    if not m:
        sys.exit("kernel not found")
    body = text[m.end():]
    body = body[:body.index(".end_amdhsa_kernel") if ".end_amdhsa_kernel" in body else len(body)]
    lines = body.splitlines()
    # the steady-state loop: from the last "AECM_MARK 0" backwards to its loop label is hard to find
    # statically; count from marker 0 to the next marker 0 / s_endpgm in textual order instead.
    idx = [i for i, l in enumerate(lines) if "AECM_MARK" in l]
    marks = [(i, int(lines[i].split("AECM_MARK")[1].split()[0])) for i in idx]
    counts = collections.OrderedDict()
    seen_first = False
    cur = None
    for i, l in enumerate(lines):
        s = l.strip()
        if "AECM_MARK" in s:
            cur = int(s.split("AECM_MARK")[1].split()[0]) + 1
            continue
        if cur is None or not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        counts.setdefault(cur, collections.Counter())[classify(op)] += 1
        counts[cur]["op:" + op] += 1
    tot = collections.Counter()
    print(f"{'phase (code after marker n-1)':44s} {'VALU':>6s} {'SALU':>6s} {'LDS':>5s} {'VMEM':>5s}")
    for ph, c in counts.items():
        print(f"{ph:2d} {PHASES.get(ph, '?'):41s} {c['VALU']:6d} {c['SALU']:6d} {c['LDS']:5d} {c['VMEM']:5d}")
        for k in ("VALU", "SALU", "LDS", "VMEM"):
            tot[k] += c[k]
    print(f"{'total (all textual code after first marker)':44s} {tot['VALU']:6d} {tot['SALU']:6d} {tot['LDS']:5d} {tot['VMEM']:5d}")
    return counts


if __name__ == "__main__":
    main()
