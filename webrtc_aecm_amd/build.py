"""Build the HIP shared library in-tree: webrtc_aecm_amd/_lib/libaecm_mi355x.so (gfx950 only), the CLI next to
it, and the precondition-audit twin libaecm_mi355x_checked.so (same sources, kernels built with -DAECM_CHECKED).

build_ubsan() is separate and opt-in (test infrastructure): the shipped build neither needs nor waits for a sanitizer
runtime."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "_lib"
LIB = LIB_DIR / "libaecm_mi355x.so"
LIB_CHECKED = LIB_DIR / "libaecm_mi355x_checked.so"
# the same kernels under host objects built with UndefinedBehaviorSanitizer (test infrastructure: tests/test_gpu_parity.py
# runs the C-ABI tests on it; never the default, built only by build_ubsan())
LIB_UBSAN = LIB_DIR / "libaecm_mi355x_ubsan.so"
UBSAN_FLAGS = ["-O1", "-g", "-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-shared-libsan", "-Wno-option-ignored"]
CLI = LIB_DIR / "aecm_run"
KERNEL_SOURCES = ["aecm_block_kernels.hip", "aecm_kernels.hip"]
HOST_SOURCES = ["aecm_engine.cpp", "aecm_session.cpp", "aecm_schedule.cpp", "aecm_sessions.cpp", "aecm_capi.cpp", "aecm_host_state.cpp"]
SOURCES = KERNEL_SOURCES + HOST_SOURCES
# Per-source flags.  The kernels branch on wave-uniform conditions only; leaving those regions unstructurized is worth +6 %
# on the block kernels and, at 7 waves per SIMD, 4 % on the tick kernel (profiles/r03_experiments.md section 7).  Two units:
# they compile in parallel, and the flags of one can be changed without the other (tools/ab_build.py name:source=flags).
UNIFORM_BRANCH_FLAGS = ["-mllvm", "-structurizecfg-skip-uniform-regions"]
# The block kernels also keep their (wave-uniform) branches as branches: no folding of two-entry phis into selects, no
# speculative hoisting out of conditional blocks -- a skipped block costs nothing, a select form executes both sides on the
# vector port that bounds the kernel (+2.7 %, section 9 of the experiment log; no effect on the tick kernel).
KEEP_BRANCHES_FLAGS = ["-mllvm", "-phi-node-folding-threshold=0", "-mllvm", "-two-entry-phi-node-folding-threshold=0",
                       "-mllvm", "-spec-exec-max-speculation-cost=0"]
SOURCE_FLAGS = {"aecm_block_kernels.hip": UNIFORM_BRANCH_FLAGS + KEEP_BRANCHES_FLAGS, "aecm_kernels.hip": UNIFORM_BRANCH_FLAGS}
# max-ilp machine scheduling measured +1.7 % on the VALU-bound block kernel (MI355X, 65 536 streams)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fwrapv", "-fPIC", "-shared",
               "-mllvm", "-amdgpu-sched-strategy=max-ilp"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm)")


def _deps():
    return list(CSRC.glob("*")) + list((PKG.parent / "include").rglob("*.h"))


def is_stale() -> bool:
    """The shipped library, its audit twin or the CLI is missing or older than a source.  The sanitizer twin is not part
    of this: a toolchain without the UBSan runtime must not make every load() retry a failing build."""
    if not LIB.exists() or not LIB_CHECKED.exists() or not CLI.exists():
        return True
    t = min(LIB.stat().st_mtime, LIB_CHECKED.stat().st_mtime)
    return any(d.stat().st_mtime > t for d in _deps())


def ubsan_is_stale() -> bool:
    if not LIB_UBSAN.exists():
        return True
    t = LIB_UBSAN.stat().st_mtime
    return any(d.stat().st_mtime > t for d in _deps())


def _compile_flags():
    return [f for f in HIPCC_FLAGS if f != "-shared"]


BUILD_INFO = LIB_DIR / "BUILD_INFO.json"


def _write_build_info():
    """Which commit the libraries were built from (the GPU box receives the tree without .git)."""
    import json
    info = {"commit": None, "dirty": None}
    try:
        root = PKG.parent
        info["commit"] = subprocess.run(["git", "-C", str(root), "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
        info["dirty"] = bool(subprocess.run(["git", "-C", str(root), "status", "--porcelain", "--", "webrtc_aecm_amd", "include"],
                                            capture_output=True, text=True).stdout.strip())
    except Exception:
        pass
    BUILD_INFO.write_text(json.dumps(info))


def _refresh_build_info():
    """The libraries are newer than every source: if the sources are exactly those of HEAD, that is the commit they were
    built from, whatever HEAD was when the compiler last ran (commits that do not touch the sources, or a commit made
    after the build)."""
    try:
        root = PKG.parent
        if not (root / ".git").exists():
            return
        dirty = subprocess.run(["git", "-C", str(root), "status", "--porcelain", "--", "webrtc_aecm_amd", "include"],
                               capture_output=True, text=True).stdout.strip()
        if not dirty:
            _write_build_info()
    except Exception:
        pass


def build_info():
    import json
    try:
        return json.loads(BUILD_INFO.read_text())
    except Exception:
        return {"commit": None, "dirty": None}


def _compile_all(hipcc, jobs, verbose):
    def compile_one(job):
        src, obj, extra = job
        flags = [f for f in _compile_flags() if not (f == "-O3" and "-O1" in extra)]
        cmd = [hipcc, *flags, *extra, "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=str(CSRC))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        list(ex.map(compile_one, jobs))


def _link(hipcc, out, objects, link_flags, verbose) -> Path:
    """Link into a temporary next to `out`; the caller moves it into place (atomically) once everything it needs exists."""
    tmp = LIB_DIR / f".{out.name}.{os.getpid()}.tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *link_flags, *objects, "-o", str(tmp)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=str(CSRC))
    return tmp


def _ubsan_runtime_dir(hipcc):
    """Directory of the shared UBSan runtime of the compiler hipcc drives (asked of that compiler; None when it has none)."""
    for clang in (Path(hipcc).resolve().parent.parent / "lib" / "llvm" / "bin" / "clang", Path(hipcc).resolve().parent / "clang",
                  Path(hipcc).resolve().parent / "amdclang"):
        try:
            if not clang.exists():
                continue
            rt = subprocess.run([str(clang), "-print-file-name=libclang_rt.ubsan_standalone-x86_64.so"],
                                capture_output=True, text=True, timeout=60).stdout.strip()
            if rt and Path(rt).is_absolute() and Path(rt).exists():
                return str(Path(rt).parent)
        except (OSError, subprocess.SubprocessError):
            continue
    return None


def build_ubsan(force: bool = False, verbose: bool = False) -> Path:
    """Test infrastructure, opt-in: libaecm_mi355x_ubsan.so = the shipped kernels under host objects compiled with
    -fsanitize=undefined.  Raises RuntimeError (and touches nothing the product loads) when the toolchain has no shared
    UBSan runtime or the build fails."""
    if not force and not ubsan_is_stale():
        return LIB_UBSAN
    import fcntl
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    with open(LIB_DIR / ".build_ubsan.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not ubsan_is_stale():
                return LIB_UBSAN
            hipcc = _hipcc()
            rt_dir = _ubsan_runtime_dir(hipcc)
            if rt_dir is None:
                raise RuntimeError("no shared UBSan runtime (libclang_rt.ubsan_standalone-x86_64.so) in hipcc's toolchain")
            obj_dir = LIB_DIR / f".ubsanobj.{os.getpid()}"      # not ".obj.*": build() sweeps those
            obj_dir.mkdir(exist_ok=True)
            try:
                jobs = [(s, obj_dir / (s + ".o"), SOURCE_FLAGS.get(s, [])) for s in KERNEL_SOURCES]
                jobs += [(s, obj_dir / (s + ".ubsan.o"), UBSAN_FLAGS) for s in HOST_SOURCES]
                _compile_all(hipcc, jobs, verbose)
                objects = [str(o) for _, o, _ in jobs]
                tmp = _link(hipcc, LIB_UBSAN, objects, ["-fsanitize=undefined", "-shared-libsan", f"-Wl,-rpath,{rt_dir}"], verbose)
                os.replace(tmp, LIB_UBSAN)
            except (OSError, subprocess.SubprocessError) as e:
                raise RuntimeError(f"UBSan build failed: {e}") from e
            finally:
                for tmp in LIB_DIR.glob(f".{LIB_UBSAN.name}.{os.getpid()}.tmp"):
                    tmp.unlink()
                shutil.rmtree(obj_dir, ignore_errors=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_UBSAN


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP/C++ source of the engine for gfx950 (one hipcc -c per source, in parallel) and link the
    shipped library, its audit twin and the CLI.

    Safe to call from several processes at once (one rank per GPU under torch.distributed.run): the
    build is serialised by a file lock and the libraries are moved into place atomically."""
    if not force and not is_stale():
        _refresh_build_info()
        return LIB
    import fcntl
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    with open(LIB_DIR / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():          # another process built it while we waited
                return LIB
            for stale in LIB_DIR.glob(".obj.*"):           # left behind by a build that was killed
                shutil.rmtree(stale, ignore_errors=True)
            obj_dir = LIB_DIR / f".obj.{os.getpid()}"
            obj_dir.mkdir(exist_ok=True)
            hipcc = _hipcc()
            jobs = [(s, obj_dir / (s + ".o"), SOURCE_FLAGS.get(s, [])) for s in SOURCES]
            jobs += [(s, obj_dir / (s + ".checked.o"), SOURCE_FLAGS.get(s, []) + ["-DAECM_CHECKED"]) for s in KERNEL_SOURCES]

            try:                                        # the object directory goes away whether or not the build succeeds
                _compile_all(hipcc, jobs, verbose)
                common = [str(obj_dir / (s + ".o")) for s in HOST_SOURCES]
                kernels = [str(obj_dir / (s + ".o")) for s in KERNEL_SOURCES]
                kernels_checked = [str(obj_dir / (s + ".checked.o")) for s in KERNEL_SOURCES]
                # both libraries are linked before either is moved into place: a failing link leaves the old pair
                tmps = [(_link(hipcc, out, kern + common, [], verbose), out)
                        for out, kern in ((LIB, kernels), (LIB_CHECKED, kernels_checked))]
                for tmp, out in tmps:
                    os.replace(tmp, out)
            finally:
                for tmp in LIB_DIR.glob(f".*.{os.getpid()}.tmp"):
                    tmp.unlink()
                shutil.rmtree(obj_dir, ignore_errors=True)
            # the command-line front end (reference main.cc equivalent + multi-file batch mode)
            tmp_cli = LIB_DIR / f".{CLI.name}.{os.getpid()}.tmp"
            cli = ["g++", "-O2", "-std=c++17", str(CSRC / "aecm_cli.cpp"), "-o", str(tmp_cli), f"-L{LIB_DIR}", "-laecm_mi355x",
                   "-pthread", "-Wl,-rpath,$ORIGIN"]
            if verbose:
                print(" ".join(cli), flush=True)
            subprocess.check_call(cli, cwd=str(CSRC))
            os.replace(tmp_cli, CLI)
            _write_build_info()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True))
    if "--ubsan" in sys.argv:
        print(build_ubsan(force=True, verbose=True))
