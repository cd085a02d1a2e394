"""Static instruction census of the kernels inside the BUILT library (diagnostics).

The block kernel is bound by instruction issue, so the numbers that explain its speed (VALU / SALU
instructions per frame, port-busy fractions; profiles/*_rocprof_summary.json) belong to one specific
binary.  This module disassembles the gfx950 code object embedded in libaecm_mi355x.so and returns, per
kernel, instruction counts by class and a fingerprint of the instruction stream; bench.py prints the
recorded issue-port figures only when the fingerprint of the library it just timed equals the one the
figures were measured on.

    python -m webrtc_aecm_amd.isa_census [lib.so]        # prints the census as JSON
"""
from __future__ import annotations

import hashlib
import json
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM_BIN = Path("/opt/rocm/lib/llvm/bin")
# The "2.3-cycle class" of the issue-rate micro-benchmark (profiles/r01_issue_port_experiments.md section 1):
# plain VOP1/VOP2 encodings of these ALU ops issue back to back in ~2.3 shader cycles per wave64 instruction,
# everything else (VOP3, shifts left, min/max, multiplies, DPP/SDWA forms, compares, selects) takes 4.
FAST_CLASS = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32",
              "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32"}
SLOW8 = {"v_permlane32_swap_b32", "v_permlane16_swap_b32", "v_sqrt_f32", "v_rcp_f32", "v_rcp_iflag_f32", "v_rsq_f32"}


def _tool(name: str) -> str:
    p = LLVM_BIN / name
    return str(p) if p.exists() else (shutil.which(name) or name)


def classify(mnemonic: str) -> str:
    op = mnemonic
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep", "s_setprio", "s_sethalt", "s_code_end")):
            return "SCTL"
        if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")):
            return "BRANCH"
        if op.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache", "s_atomic")):
            return "SMEM"
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    return "other"


def _base(mnemonic: str):
    """(base opcode, encoding suffix) of a disassembled VALU mnemonic, e.g. v_add_u32_e32 -> (v_add_u32, e32)."""
    m = re.match(r"^(.*?)_(e32|e64|dpp|sdwa)$", mnemonic)
    return (m.group(1), m.group(2)) if m else (mnemonic, "")


def disassemble(lib_path) -> str:
    """Disassembly text of the gfx950 code objects bundled in the shared library (one per kernel translation unit)."""
    lib_path = Path(lib_path)
    with tempfile.TemporaryDirectory() as td:
        tmp = Path(td) / lib_path.name
        shutil.copy(lib_path, tmp)                                   # --offloading writes next to its input
        subprocess.run([_tool("llvm-objdump"), "--offloading", str(tmp)], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        objs = sorted(Path(td).glob(tmp.name + ".*amdgcn*gfx950*"))
        if not objs:
            raise RuntimeError(f"no gfx950 code object found in {lib_path}")
        return "\n".join(subprocess.run([_tool("llvm-objdump"), "-d", str(o)], check=True, capture_output=True, text=True).stdout
                         for o in objs)


def census_of_text(text: str):
    """{kernel symbol: {class counts, fast-class count, 8-cycle count, opcode histogram, fingerprint}}."""
    out = {}
    cur, ops = None, []

    def close():
        if cur is None or not ops:
            return
        c = {"VALU": 0, "SALU": 0, "SCTL": 0, "BRANCH": 0, "SMEM": 0, "LDS": 0, "VMEM": 0, "other": 0}
        hist, fast, slow8 = {}, 0, 0
        for mn, _ in ops:
            c[classify(mn)] += 1
            hist[mn] = hist.get(mn, 0) + 1
            b, enc = _base(mn)
            if b in FAST_CLASS and enc in ("e32", ""):
                fast += 1
            if b in SLOW8:
                slow8 += 1
        fp = hashlib.sha256("\n".join(f"{mn} {args}" for mn, args in ops).encode()).hexdigest()[:16]
        out[cur] = dict(counts=c, valu_fast_class=fast, valu_8cycle_class=slow8, n_instructions=len(ops), fingerprint=fp,
                        opcodes=dict(sorted(hist.items(), key=lambda kv: -kv[1])))

    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
        if m:
            close()
            cur, ops = m.group(1), []
            continue
        if cur is None:
            continue
        s = line.strip()
        if not s or s.startswith(("//", ";")):
            continue
        body = s.split("//")[0].strip()
        if not body:
            continue
        parts = body.split(None, 1)
        mn = parts[0]
        if not re.match(r"^[a-z_0-9]+$", mn):
            continue
        args = parts[1] if len(parts) > 1 else ""
        args = re.sub(r"<[^>]*>", "", args).strip()                  # symbolic branch targets carry the symbol name
        ops.append((mn, args))
    close()
    return out


# The block kernel a launch takes, by WebRtcAecmBatch_DescribeLaunch's form (fast variant; with / without a clean input):
# mangled-name fragment and the name bench.py prints.
BLOCK_KERNELS = {
    (0, False): ("aecm_process_kernelILb1ELb0ELb0", "aecm_process_kernel<fast,noclean,rotation>"),
    (0, True): ("aecm_process_kernelILb1ELb1ELb0", "aecm_process_kernel<fast,clean,rotation>"),
    (1, False): ("aecm_process_kernelILb1ELb0ELb1", "aecm_process_kernel<fast,noclean>"),
    (1, True): ("aecm_process_kernelILb1ELb1ELb1", "aecm_process_kernel<fast,clean>"),
    (2, False): ("aecm_process_queue_kernelILb0E", "aecm_process_queue_kernel<noclean>"),
    (2, True): ("aecm_process_queue_kernelILb1E", "aecm_process_queue_kernel<clean>"),
    (3, False): ("aecm_process_pipelined_kernelILi0ELb1ELb1ELi2E", "aecm_process_pipelined_kernel<tail=0,front=2,raw,balance>"),
}


def block_kernel(form: int, clean: bool, detail: int = 0):
    """(mangled-name fragment, printed name) of the kernel a launch takes; detail = DescribeLaunch's second value (for the
    pipelined form: the tail waves per workgroup + 0x100 balanced + 0x200 four front waves + 0x400 raw hand-over + 0x800 delay
    waves + 0x1000 gain waves, the kernel's template arguments).  The fragment is a regular expression."""
    if form == 3:
        tail, bal, front, raw = detail & 0xff, (detail >> 8) & 1, 4 if detail & 0x200 else 2, (detail >> 10) & 1
        gain = 4 if detail & 0x1000 else 0
        delay = (2 if gain else 4) if detail & 0x800 else 0
        return (f"aecm_process_pipelined_kernelILi{tail}ELb{bal}ELb{raw}ELi{front}ELi{delay}ELi{gain}E",
                f"aecm_process_pipelined_kernel<tail={tail},front={front}{f',delay={delay}' if delay else ''}{f',gain={gain}' if gain else ''}"
                f"{',raw' if raw else ''}{',balance' if bal else ''}>")
    return BLOCK_KERNELS[(form, clean)]
HEADLINE_KERNEL = BLOCK_KERNELS[(2, False)][0]       # bench.py's default workload (65 536 streams: larger than the chip)


def census(lib_path, kernel_substr: str = HEADLINE_KERNEL):
    """Census of the first kernel whose mangled name matches kernel_substr (a regular expression; default: the headline block kernel)."""
    all_k = census_of_text(disassemble(lib_path))
    for name, c in all_k.items():
        if re.search(kernel_substr, name):
            return dict(kernel=name, **c)
    raise KeyError(kernel_substr)


if __name__ == "__main__":
    from . import build as _build
    lib = sys.argv[1] if len(sys.argv) > 1 else _build.LIB
    c = census(lib, sys.argv[2] if len(sys.argv) > 2 else HEADLINE_KERNEL)
    c["opcodes"] = dict(list(c["opcodes"].items())[:40])
    print(json.dumps(c, indent=1))
