#!/usr/bin/env python3
"""Static instruction census of the block kernel by phase.

Compiles csrc/aecm_block_kernels.hip (with its build flags) for gfx950 with -DAECM_MARKERS (comment markers at the phase
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
FAST_CLASS = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32",
              "v_ashrrev_i32", "v_mov_b32"}          # the ~2.3-cycle class of the issue-rate micro-benchmark (VGPR / constant operands only)
SLOW8 = {"v_permlane32_swap_b32", "v_permlane16_swap_b32", "v_sqrt_f32"}
PHASES = {0: "load+prefetch", 1: "far window+FFT+mag", 2: "near window+FFT+mag", 3: "far history/binary far",
          4: "binary near + delay estimator", 5: "aligned far fetch", 6: "energies/VAD", 7: "NLMS channel update",
          8: "store/restore + supgain + Wiener", 9: "NLP + comfort noise prep", 10: "comfort noise",
          11: "IFFT", 12: "synthesis + store", 13: "output store + loop control", 14: "(after the loop: state store)", 15: "(epilogue)"}


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
    ap.add_argument("--kernel", default="aecm_process_kernelILb1ELb0ELb1EE")
    ap.add_argument("--dump")
    ap.add_argument("--extra", default="", help="extra hipcc flags")
    ap.add_argument("--hot", action="store_true",
                    help="-DAECM_CENSUS_HOTPATH: steady-state-unreachable branches compiled out, so the block loop is the straight-line "
                         "hot path and its static count approximates the dynamic mix; also prints the fast-class share per phase")
    ap.add_argument("--json", help="write the per-phase counts here")
    ap.add_argument("--runs", action="store_true", help="also print how the fast-class instructions of the block loop are distributed "
                    "over uninterrupted runs (a run ends at any other instruction of the wave: slower VALU, SALU, LDS, memory, wait)")
    a = ap.parse_args()
    sys.path.insert(0, str(ROOT))
    from webrtc_aecm_amd import build as B
    flags = [f for f in B.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] if hasattr(B, "HIPCC_FLAGS") else \
        ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fwrapv", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "k.s"
        src = "aecm_block_kernels.hip"
        cmd = ["/opt/rocm/bin/hipcc", *flags, *B.SOURCE_FLAGS.get(src, []), *a.extra.split(), *(["-DAECM_CENSUS_HOTPATH"] if a.hot else []), "-DAECM_MARKERS",
               "-S", "--cuda-device-only", f"-I{CSRC}", str(CSRC / src), "-o", str(out)]
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
    runs = collections.Counter()          # length of a run of fast-class instructions -> how many such runs (phases 1..13)
    run = 0
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
        base = op
        for suf in ("_e32", "_e64"):
            if base.endswith(suf):
                base = base[:-4]
        operands = s.split(None, 1)[1] if len(s.split(None, 1)) > 1 else ""
        srcs = [x.strip() for x in operands.split(",")[1:]]
        if base in FAST_CLASS and not op.endswith(("_dpp", "_sdwa")) and not any(x.startswith(("s", "vcc", "exec", "m0", "ttmp")) for x in srcs):
            counts[cur]["VALU_fast"] += 1
            run += 1 if cur <= 13 else 0
        else:
            if run:
                runs[run] += 1
            run = 0
        if base in SLOW8:
            counts[cur]["VALU_8cyc"] += 1
    tot = collections.Counter()
    print(f"{'phase (code after marker n-1)':44s} {'VALU':>6s} {'fast':>5s} {'8cyc':>5s} {'SALU':>6s} {'SCTL':>5s} {'LDS':>5s} {'VMEM':>5s}")
    for ph, c in counts.items():
        print(f"{ph:2d} {PHASES.get(ph, '?'):41s} {c['VALU']:6d} {c['VALU_fast']:5d} {c['VALU_8cyc']:5d} {c['SALU']:6d} {c['SCTL']:5d} {c['LDS']:5d} {c['VMEM']:5d}")
        if ph <= 13:
            for k in ("VALU", "VALU_fast", "VALU_8cyc", "SALU", "SCTL", "LDS", "VMEM"):
                tot[k] += c[k]
    print(f"{'block loop (phases 1..13)':44s} {tot['VALU']:6d} {tot['VALU_fast']:5d} {tot['VALU_8cyc']:5d} {tot['SALU']:6d} {tot['SCTL']:5d} {tot['LDS']:5d} {tot['VMEM']:5d}")
    if a.runs:
        n = sum(k * v for k, v in runs.items())
        print("fast-class instructions by length of their uninterrupted run: " +
              ", ".join(f"{k}: {v} runs" for k, v in sorted(runs.items())) +
              f"; in runs of >= 4: {sum(k * v for k, v in runs.items() if k >= 4)} of {n}, >= 7: {sum(k * v for k, v in runs.items() if k >= 7)}")
    if a.json:
        import json
        Path(a.json).write_text(json.dumps({"hot": a.hot, "kernel": a.kernel, "loop_total": dict(tot),
                                            "phases": {str(ph): {"name": PHASES.get(ph, "?"), **{k: v for k, v in c.items() if not k.startswith("op:")},
                                                                 "top_ops": dict(collections.Counter({k[3:]: v for k, v in c.items() if k.startswith("op:")}).most_common(12))}
                                                       for ph, c in counts.items()}}, indent=1))
    return counts


if __name__ == "__main__":
    main()
