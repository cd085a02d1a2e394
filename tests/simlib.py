"""TEST INFRASTRUCTURE: build + bind tests/_build/libaecm_sim.so -- the product's wave-generic block
DSP (webrtc_aecm_amd/csrc/aecm_wave.h) instantiated on a 64-lane CPU simulator (tests/sim/)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "webrtc_aecm_amd" / "csrc"
# AECM_SIM_SANITIZE=1: the same sources built with AddressSanitizer + UndefinedBehaviorSanitizer into their own library
# (tests/test_sanitizers.py runs part of the CPU suite on it in a child process with the sanitizer runtimes preloaded)
SANITIZE = os.environ.get("AECM_SIM_SANITIZE") == "1"
SIM_SO = ROOT / "tests" / "_build" / ("libaecm_sim_san.so" if SANITIZE else "libaecm_sim.so")
SAN_FLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
_SOURCES = [ROOT / "tests" / "sim" / "sim_lib.cpp", ROOT / "tests" / "sim" / "sim_engine.cpp", ROOT / "tests" / "sim" / "sim_flow.cpp",
            CSRC / "aecm_host_state.cpp", CSRC / "aecm_session.cpp", CSRC / "aecm_schedule.cpp"]
_DEPS = _SOURCES + [ROOT / "tests" / "sim" / "wave_sim.h", CSRC / "aecm_wave.h", CSRC / "aecm_ops.h",
                    CSRC / "aecm_state.h", CSRC / "aecm_host_state.h", CSRC / "aecm_tables.h",
                    CSRC / "aecm_session.h", CSRC / "aecm_engine.h", CSRC / "aecm_session_flow.h", CSRC / "aecm_flow_plan.h"]
_i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_lib = None


def build():
    if SIM_SO.exists() and all(SIM_SO.stat().st_mtime >= d.stat().st_mtime for d in _DEPS):
        return
    SIM_SO.parent.mkdir(parents=True, exist_ok=True)
    # one object per source, compiled in parallel (the block DSP template is instantiated in three of them; the sanitizer
    # build of a single g++ call took more than two minutes)
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = SIM_SO.parent / (".obj_san" if SANITIZE else ".obj")
    obj_dir.mkdir(exist_ok=True)
    flags = [*(SAN_FLAGS if SANITIZE else ["-O2"]), "-std=c++17", "-fwrapv", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
             f"-I{CSRC}", f"-I{ROOT / 'tests' / 'sim'}"]

    def compile_one(src):
        obj = obj_dir / (Path(src).name + ".o")
        subprocess.check_call(["g++", *flags, "-c", str(src), "-o", str(obj)])
        return str(obj)
    with ThreadPoolExecutor(max_workers=min(len(_SOURCES), os.cpu_count() or 2)) as ex:
        objs = list(ex.map(compile_one, _SOURCES))
    tmp = SIM_SO.with_suffix(f".{os.getpid()}.tmp")
    subprocess.check_call(["g++", *(["-fsanitize=address,undefined"] if SANITIZE else []), "-shared", *objs, "-o", str(tmp)])
    os.replace(tmp, SIM_SO)


def lib():
    global _lib
    if _lib is None:
        build()
        l = C.CDLL(str(SIM_SO))
        l.sim_create.restype = C.c_void_p
        l.sim_create.argtypes = [C.c_int, C.c_int, C.c_int]
        l.sim_free.argtypes = [C.c_void_p]
        l.sim_control.argtypes = [C.c_void_p, C.c_int, C.c_int]
        l.sim_set_echo_path.argtypes = [C.c_void_p, _i16p]
        l.sim_get_echo_path.argtypes = [C.c_void_p, _i16p]
        l.sim_process.argtypes = [C.c_void_p, _i16p, _i16p, C.c_void_p, _i16p, C.c_int]
        l.sim_digest.argtypes = [C.c_void_p, _u32p]
        l.sim_process_roles.argtypes = [C.c_void_p, _i16p, _i16p, _i16p, C.c_int, C.c_int]
        l.sim_fft128.argtypes = [_i16p, _i16p, C.c_int]
        l.sim_constants.argtypes = [_u32p, _u32p, _u32p]
        l.sim_recordings.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
        l.simsession_create.restype = C.c_void_p
        l.simsession_free.argtypes = [C.c_void_p]
        l.simsession_init.argtypes = [C.c_void_p, C.c_int32]
        l.simsession_buffer_farend.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        l.simsession_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int16]
        l.simsession_set_config.argtypes = [C.c_void_p, C.c_int16, C.c_int16]
        l.simsession_init_echo_path.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        l.simsession_get_echo_path.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        l.sim_flow_fuzz.restype = C.c_int64
        l.sim_flow_fuzz.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_int64)]
        l.sim_flow_tolerance_check.restype = C.c_int
        l.sim_div_magic_check.restype = C.c_int
        _lib = l
    return _lib


def flow_fuzz(seed, fs, n_ticks, scenario, start_pos=0):
    """(first differing tick or -1, [what, blocks processed, ticks past start-up, ticks with dropped far samples,
    ticks with direct far fetches, replay frames moved to their rows, far-end calls outside ticks, samples of those dropped])."""
    detail = (C.c_int64 * 8)()
    tick = lib().sim_flow_fuzz(seed, fs, n_ticks, scenario, start_pos, detail)
    return tick, list(detail)


class SimStream:
    def __init__(self, fs=16000, cng_mode=1, echo_mode=3):
        self.lib = lib()
        self.h = self.lib.sim_create(fs, cng_mode, echo_mode)
        if not self.h:
            raise ValueError("bad parameters")

    def control(self, fixed_delay, nlp_flag):
        self.lib.sim_control(self.h, fixed_delay, nlp_flag)

    def process(self, far, near, clean=None):
        far = np.ascontiguousarray(far, dtype=np.int16)
        near = np.ascontiguousarray(near, dtype=np.int16)
        out = np.empty_like(near)
        cptr = None
        if clean is not None:
            clean = np.ascontiguousarray(clean, dtype=np.int16)
            cptr = clean.ctypes.data_as(C.c_void_p)
        self.lib.sim_process(self.h, far, near, cptr, out, far.size // 64)
        return out

    def process_roles(self, far, near, order=0):
        """The same blocks through the pipelined kernel's role decomposition (tests/sim/sim_lib.cpp: sim_process_roles)."""
        far = np.ascontiguousarray(far, dtype=np.int16)
        near = np.ascontiguousarray(near, dtype=np.int16)
        out = np.empty_like(near)
        self.lib.sim_process_roles(self.h, far, near, out, far.size // 64, order)
        return out

    def digest(self):
        d = np.zeros(24, dtype=np.uint32)
        self.lib.sim_digest(self.h, d)
        return d

    def init_echo_path(self, path):
        self.lib.sim_set_echo_path(self.h, np.ascontiguousarray(path, dtype=np.int16))

    def echo_path(self):
        p = np.zeros(65, dtype=np.int16)
        self.lib.sim_get_echo_path(self.h, p)
        return p

    def __del__(self):
        try:
            self.lib.sim_free(self.h)
        except Exception:
            pass


class SimSession:
    """The product's Session class (host logic) over the simulated engine, ABI-shaped."""

    def __init__(self):
        self.lib = lib()
        self.h = self.lib.simsession_create()

    def init(self, fs):
        return self.lib.simsession_init(self.h, fs)

    def set_config(self, cng, em):
        return self.lib.simsession_set_config(self.h, cng, em)

    def buffer_farend(self, far):
        far = np.ascontiguousarray(far, dtype=np.int16)
        return self.lib.simsession_buffer_farend(self.h, far.ctypes.data, far.size)

    def process(self, near, clean=None, ms=0):
        near = np.ascontiguousarray(near, dtype=np.int16)
        out = np.empty_like(near)
        cp = None
        if clean is not None:
            clean = np.ascontiguousarray(clean, dtype=np.int16)
            cp = clean.ctypes.data
        rc = self.lib.simsession_process(self.h, near.ctypes.data, cp, out.ctypes.data, near.size, ms)
        return rc, out

    def init_echo_path(self, path):
        path = np.ascontiguousarray(path, dtype=np.int16)
        return self.lib.simsession_init_echo_path(self.h, path.ctypes.data, path.nbytes)

    def get_echo_path(self):
        out = np.zeros(65, dtype=np.int16)
        return self.lib.simsession_get_echo_path(self.h, out.ctypes.data, out.nbytes), out

    def __del__(self):
        try:
            self.lib.simsession_free(self.h)
        except Exception:
            pass


def sim_recordings(far, near, fs, frame, cng, echo_mode, ms, clean=None):
    """Schedule-based batched sessions on the simulator: returns (code, out) like
    AecmBatch.process_recordings_host."""
    far = np.ascontiguousarray(far, dtype=np.int16)
    near = np.ascontiguousarray(near, dtype=np.int16)
    cptr = None
    if clean is not None:
        clean = np.ascontiguousarray(clean, dtype=np.int16)
        cptr = clean.ctypes.data
    out = near.copy()
    rc = lib().sim_recordings(far.shape[0], far.shape[1], fs, frame, cng, echo_mode, ms, far.ctypes.data, near.ctypes.data,
                              cptr, out.ctypes.data)
    return rc, out


N_LANE_CONST_ROWS = 12          # aecm_state.h: kLaneConstRows


def constants():
    """(host-built blob, lane-constant rows from their definitions, twiddle pairs from their definitions)."""
    blob = np.zeros(N_LANE_CONST_ROWS * 64 + 7 * 64 * 4 + 6 * 64 * 4 + 3 * 64 * 4 + 360 + 68, dtype=np.uint32)
    rows = np.zeros(N_LANE_CONST_ROWS * 64, dtype=np.uint32)
    tw = np.zeros(7 * 64 * 4 + 6 * 64 * 4 + 3 * 64 * 4, dtype=np.uint32)
    lib().sim_constants(blob, rows, tw)
    return blob, rows, tw
