"""TEST INFRASTRUCTURE -- ctypes bindings for the CPU oracle and (when present) the real reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

  Oracle        : oracle/_build/libaecm_oracle.so  (our plain-C restatement, oracle/aecm_oracle.c)
  Reference     : oracle/_ref/libaecm_ref.so       (the unmodified reference + oracle/ref_shim.cc;
                  built in the container that has /root/reference, travels prebuilt elsewhere)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ORACLE_SO = HERE / "_build" / "libaecm_oracle.so"
REF_SO = HERE / "_ref" / "libaecm_ref.so"
REFMAIN = HERE / "_ref" / "aecm_run_refmain"     # the reference's main.cc, unmodified, linked against the MI355X library
REFWAV = HERE / "_ref" / "ref_wavdec"            # the WAV reader the reference CLI uses (dr_wav.h) behind oracle/ref_wavdec.cc
BLOCK = 64
BINS = 65
DIGEST_WORDS = 24

DIGEST_NAMES = [
    "totCount", "seed", "startupState|far_history_pos", "dfaNoisyQ|Old", "farLog|farEnergyMin",
    "farEnergyMax|MaxMin", "farEnergyVAD|MSE", "currentVAD|vadUpdateCount", "firstVAD|mseChannelCount",
    "mseAdaptOld", "mseStoredOld", "mseThreshold", "supGain|supGainOld", "last_delay",
    "minimum_probability", "last_delay_probability", "H(chStored,chAdapt16)", "H(chAdapt32)",
    "H(echoFilt)", "H(nearFilt)", "H(noiseEst,ctrs)", "H(delay estimator)", "H(log energies)",
    "H(xBuf,dBuf,outBuf,far_history)",
]

_i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> None:
    """Compile the oracle (always possible: gcc) and the reference (only if /root/reference exists)."""
    if force or not ORACLE_SO.exists() or ORACLE_SO.stat().st_mtime < (HERE / "aecm_oracle.c").stat().st_mtime:
        subprocess.check_call(["make", "-C", str(HERE), "oracle"], stdout=subprocess.DEVNULL)
    if Path("/root/reference/aecm").is_dir() and (force or not REF_SO.exists()):
        subprocess.check_call(["make", "-C", str(HERE), "ref"], stdout=subprocess.DEVNULL)
    if Path("/root/reference/dr_wav.h").is_file() and (force or not REFWAV.exists() or REFWAV.stat().st_mtime < (HERE / "ref_wavdec.cc").stat().st_mtime):
        subprocess.run(["make", "-C", str(HERE), "refwav"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)   # best effort, like refmain
    product = HERE.parent / "webrtc_aecm_amd" / "_lib" / "libaecm_mi355x.so"
    if Path("/root/reference/main.cc").is_file() and product.exists() and (
            force or not REFMAIN.exists() or REFMAIN.stat().st_mtime < product.stat().st_mtime):
        # best effort: only the one test that runs the reference's main.cc needs it (it skips without the binary); a
        # link problem here must not fail the pure-oracle tests
        r = subprocess.run(["make", "-C", str(HERE), "refmain"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            import sys
            print(f"pyoracle: building oracle/_ref/aecm_run_refmain failed (test infrastructure only):\n{r.stderr[-800:]}", file=sys.stderr)
            if REFMAIN.exists() and REFMAIN.stat().st_mtime < product.stat().st_mtime:
                REFMAIN.unlink()                  # never leave a binary linked against an older library interface


def have_reference() -> bool:
    return REF_SO.exists()


_oracle_lib = None
_ref_lib = None


def oracle_lib():
    global _oracle_lib
    if _oracle_lib is None:
        build()
        lib = C.CDLL(str(ORACLE_SO))
        lib.aecm_oracle_create.restype = C.c_void_p
        lib.aecm_oracle_free.argtypes = [C.c_void_p]
        lib.aecm_oracle_init.argtypes = [C.c_void_p, C.c_int]
        lib.aecm_oracle_set_config.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.aecm_oracle_control.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.aecm_oracle_init_echo_path.argtypes = [C.c_void_p, _i16p]
        lib.aecm_oracle_get_echo_path.argtypes = [C.c_void_p, _i16p]
        lib.aecm_oracle_process_block.argtypes = [C.c_void_p, _i16p, _i16p, C.c_void_p, _i16p]
        lib.aecm_oracle_process_stream.argtypes = [C.c_void_p, _i16p, _i16p, _i16p, C.c_size_t]
        lib.aecm_oracle_digest.argtypes = [C.c_void_p, _u32p]
        lib.aecm_oracle_get_stats.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")]
        lib.aecm_oracle_sqrt_floor.argtypes = [C.c_int32]
        lib.aecm_oracle_sqrt_floor.restype = C.c_int32
        lib.aecm_oracle_fft128.argtypes = [_i16p, _i16p, C.c_int, C.POINTER(C.c_int)]
        _oracle_lib = lib
    return _oracle_lib


def ref_lib():
    global _ref_lib
    if _ref_lib is None:
        build()
        if not REF_SO.exists():
            raise FileNotFoundError(f"{REF_SO} not built (needs /root/reference)")
        lib = C.CDLL(str(REF_SO))
        lib.refshim_core_create.restype = C.c_void_p
        lib.refshim_core_init.argtypes = [C.c_void_p, C.c_int]
        lib.refshim_core_free.argtypes = [C.c_void_p]
        lib.refshim_core_process_block.argtypes = [C.c_void_p, _i16p, _i16p, C.c_void_p, _i16p]
        lib.refshim_core_init_echo_path.argtypes = [C.c_void_p, _i16p]
        lib.refshim_core_control.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.refshim_core_set_config.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.refshim_core_digest.argtypes = [C.c_void_p, _u32p]
        lib.refshim_core_process_stream.argtypes = [C.c_void_p, _i16p, _i16p, _i16p, C.c_size_t]
        # public session ABI (reference aecm/echo_control_mobile.h:46-202)
        lib.WebRtcAecm_Create.restype = C.c_void_p
        lib.WebRtcAecm_Free.argtypes = [C.c_void_p]
        lib.WebRtcAecm_Init.argtypes = [C.c_void_p, C.c_int32]
        lib.WebRtcAecm_BufferFarend.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.WebRtcAecm_GetBufferFarendError.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.WebRtcAecm_Process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int16]
        lib.WebRtcAecm_set_config.argtypes = [C.c_void_p, AecmConfig]
        lib.WebRtcAecm_InitEchoPath.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.WebRtcAecm_GetEchoPath.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.WebRtcAecm_echo_path_size_bytes.restype = C.c_size_t
        _ref_lib = lib
    return _ref_lib


class AecmConfig(C.Structure):
    _fields_ = [("cngMode", C.c_int16), ("echoMode", C.c_int16)]


class OracleStream:
    """One AECM block stream on the CPU oracle."""

    def __init__(self, fs=16000, cng_mode=1, echo_mode=3):
        self.lib = oracle_lib()
        self.h = self.lib.aecm_oracle_create()
        if self.lib.aecm_oracle_init(self.h, fs) != 0:
            raise ValueError("bad sample rate")
        if self.lib.aecm_oracle_set_config(self.h, cng_mode, echo_mode) != 0:
            raise ValueError("bad config")

    def control(self, fixed_delay, nlp_flag):
        self.lib.aecm_oracle_control(self.h, fixed_delay, nlp_flag)

    def process(self, far, near):
        far = np.ascontiguousarray(far, dtype=np.int16)
        near = np.ascontiguousarray(near, dtype=np.int16)
        out = np.empty_like(near)
        assert far.size % BLOCK == 0 and far.size == near.size
        self.lib.aecm_oracle_process_stream(self.h, far, near, out, far.size // BLOCK)
        return out

    def process_block_clean(self, far, near, clean):
        out = np.empty(BLOCK, dtype=np.int16)
        clean = np.ascontiguousarray(clean, dtype=np.int16)
        self.lib.aecm_oracle_process_block(self.h, np.ascontiguousarray(far), np.ascontiguousarray(near),
                                           clean.ctypes.data_as(C.c_void_p), out)
        return out

    def digest(self):
        d = np.zeros(DIGEST_WORDS, dtype=np.uint32)
        self.lib.aecm_oracle_digest(self.h, d)
        return d

    def stats(self):
        """Branch statistics since init (oracle/aecm_oracle.h: ORC_STAT_*) as a dict of block counts."""
        st = np.zeros(8, dtype=np.uint64)
        self.lib.aecm_oracle_get_stats(self.h, st)
        names = ("blocks", "nlms", "gain_zero", "q_steady", "ifft_unscaled", "delayed", "vad")
        return {n: int(st[i]) for i, n in enumerate(names)}

    def echo_path(self):
        p = np.zeros(BINS, dtype=np.int16)
        self.lib.aecm_oracle_get_echo_path(self.h, p)
        return p

    def init_echo_path(self, path):
        self.lib.aecm_oracle_init_echo_path(self.h, np.ascontiguousarray(path, dtype=np.int16))

    def __del__(self):
        try:
            self.lib.aecm_oracle_free(self.h)
        except Exception:
            pass


class RefCoreStream:
    """One AECM block stream on the real reference core (WebRtcAecm_ProcessBlock driven directly)."""

    def __init__(self, fs=16000, cng_mode=1, echo_mode=3):
        self.lib = ref_lib()
        self.h = self.lib.refshim_core_create()
        if self.lib.refshim_core_init(self.h, fs) != 0:
            raise ValueError("bad sample rate")
        if self.lib.refshim_core_set_config(self.h, cng_mode, echo_mode) != 0:
            raise ValueError("bad config")

    def control(self, fixed_delay, nlp_flag):
        self.lib.refshim_core_control(self.h, fixed_delay, nlp_flag)

    def process(self, far, near):
        far = np.ascontiguousarray(far, dtype=np.int16)
        near = np.ascontiguousarray(near, dtype=np.int16)
        out = np.empty_like(near)
        self.lib.refshim_core_process_stream(self.h, far, near, out, far.size // BLOCK)
        return out

    def process_block_clean(self, far, near, clean):
        out = np.empty(BLOCK, dtype=np.int16)
        clean = np.ascontiguousarray(clean, dtype=np.int16)
        self.lib.refshim_core_process_block(self.h, np.ascontiguousarray(far), np.ascontiguousarray(near),
                                         clean.ctypes.data_as(C.c_void_p), out)
        return out

    def digest(self):
        d = np.zeros(DIGEST_WORDS, dtype=np.uint32)
        self.lib.refshim_core_digest(self.h, d)
        return d

    def init_echo_path(self, path):
        self.lib.refshim_core_init_echo_path(self.h, np.ascontiguousarray(path, dtype=np.int16))

    def __del__(self):
        try:
            self.lib.refshim_core_free(self.h)
        except Exception:
            pass


class RefSession:
    """The reference's public session ABI (WebRtcAecm_Create/Init/BufferFarend/Process)."""

    def __init__(self, fs=16000, cng_mode=1, echo_mode=3):
        self.lib = ref_lib()
        self.h = self.lib.WebRtcAecm_Create()
        rc = self.lib.WebRtcAecm_Init(self.h, fs)
        if rc != 0:
            raise ValueError(rc)
        rc = self.lib.WebRtcAecm_set_config(self.h, AecmConfig(cng_mode, echo_mode))
        if rc != 0:
            raise ValueError(rc)

    # one-call doors, same shapes as webrtc_aecm_amd.Aecm (tests drive both with the same sequence)
    def init(self, fs):
        return self.lib.WebRtcAecm_Init(self.h, fs)

    def set_config(self, cng_mode, echo_mode):
        return self.lib.WebRtcAecm_set_config(self.h, AecmConfig(cng_mode, echo_mode))

    def buffer_farend(self, far):
        far = np.ascontiguousarray(far, dtype=np.int16)
        return self.lib.WebRtcAecm_BufferFarend(self.h, far.ctypes.data, far.size)

    def process(self, near, clean=None, ms=0):
        near = np.ascontiguousarray(near, dtype=np.int16)
        out = np.empty_like(near)
        cp = None
        if clean is not None:
            clean = np.ascontiguousarray(clean, dtype=np.int16)
            cp = clean.ctypes.data
        rc = self.lib.WebRtcAecm_Process(self.h, near.ctypes.data, cp, out.ctypes.data, near.size, ms)
        return rc, out

    def init_echo_path(self, path):
        path = np.ascontiguousarray(path, dtype=np.int16)
        return self.lib.WebRtcAecm_InitEchoPath(self.h, path.ctypes.data, path.nbytes)

    def get_echo_path(self):
        out = np.zeros(BINS, dtype=np.int16)
        return self.lib.WebRtcAecm_GetEchoPath(self.h, out.ctypes.data, out.nbytes), out

    def run(self, far, near, frame, ms=40):
        """main.cc:105-143 loop: BufferFarend + Process per `frame` samples; returns processed near."""
        far = np.ascontiguousarray(far, dtype=np.int16)
        near = np.ascontiguousarray(near, dtype=np.int16).copy()
        out = np.empty(frame, dtype=np.int16)
        for i in range(near.size // frame):
            f = far[i * frame:(i + 1) * frame]
            d = near[i * frame:(i + 1) * frame]
            rc = self.lib.WebRtcAecm_BufferFarend(self.h, f.ctypes.data, frame)
            assert rc == 0, rc
            rc = self.lib.WebRtcAecm_Process(self.h, d.ctypes.data, None, out.ctypes.data, frame, ms)
            assert rc == 0, rc
            d[:] = out
        return near

    def __del__(self):
        try:
            self.lib.WebRtcAecm_Free(self.h)
        except Exception:
            pass
