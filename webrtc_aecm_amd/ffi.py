"""ctypes bindings of libaecm_mi355x.so.

`Aecm` mirrors the reference's session interface (aecm/echo_control_mobile.h: Create / Init /
BufferFarend / Process / set_config / InitEchoPath / GetEchoPath / Free, same argument meaning and
return codes); `AecmBatch` is the batch extension (include/aecm_batch.h).  There is no Python or CPU
implementation of the DSP here: without the HIP library and a GPU these classes raise.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

from . import build as _build

BLOCK = 64
BINS = 65
DIGEST_WORDS = 24
KERNEL_SAFE, KERNEL_FAST = 0, 1

AECM_UNSPECIFIED_ERROR = 12000
AECM_UNSUPPORTED_FUNCTION_ERROR = 12001
AECM_UNINITIALIZED_ERROR = 12002
AECM_NULL_POINTER_ERROR = 12003
AECM_BAD_PARAMETER_ERROR = 12004
AECM_BAD_PARAMETER_WARNING = 12100

SESSION_SYMBOLS = [
    "WebRtcAecm_Create", "WebRtcAecm_Free", "WebRtcAecm_Init", "WebRtcAecm_BufferFarend",
    "WebRtcAecm_GetBufferFarendError", "WebRtcAecm_Process", "WebRtcAecm_set_config",
    "WebRtcAecm_InitEchoPath", "WebRtcAecm_GetEchoPath", "WebRtcAecm_echo_path_size_bytes",
]
BATCH_SYMBOLS = [
    "WebRtcAecmBatch_Create", "WebRtcAecmBatch_Free", "WebRtcAecmBatch_num_streams", "WebRtcAecmBatch_Init",
    "WebRtcAecmBatch_set_config", "WebRtcAecmBatch_Control", "WebRtcAecmBatch_ProcessBlocks",
    "WebRtcAecmBatch_ProcessBlocksHost", "WebRtcAecmBatch_ProcessRecordings", "WebRtcAecmBatch_ProcessRecordingsHost",
    "WebRtcAecmBatch_Synchronize", "WebRtcAecmBatch_GetLastLaunchMs",
    "WebRtcAecmBatch_GetTimers", "WebRtcAecmBatch_ResetTimers", "WebRtcAecmBatch_InitEchoPath",
    "WebRtcAecmBatch_GetEchoPath", "WebRtcAecmBatch_state_size_bytes", "WebRtcAecmBatch_ExportState",
    "WebRtcAecmBatch_ImportState", "WebRtcAecmBatch_ExportStates", "WebRtcAecmBatch_ImportStates", "WebRtcAecmBatch_ExportStatesDevice",
    "WebRtcAecmBatch_ImportStatesDevice", "WebRtcAecmBatch_GetDigest", "WebRtcAecmBatch_SetKernelVariant", "WebRtcAecmBatch_SetLaunchChunking", "WebRtcAecmBatch_SetLaunchPipelining", "WebRtcAecmBatch_DescribeLaunch", "WebRtcAecmBatch_DescribeLaunchFor",
    "WebRtcAecmBatch_SelfTest", "WebRtcAecmBatch_DebugFft128", "WebRtcAecmBatch_DeviceInfo", "WebRtcAecmBatch_GetCheckCounters",
    "WebRtcAecmBatch_RegisterHostBuffer", "WebRtcAecmBatch_UnregisterHostBuffer",
    "WebRtcAecmBatch_DefaultLaunchPolicy", "WebRtcAecmBatch_GetLaunchPolicy", "WebRtcAecmBatch_SetLaunchPolicy", "WebRtcAecmBatch_DescribeLaunchDetail",
    "WebRtcAecm_SetDefaultDevice", "WebRtcAecmBatch_DevicePciBusId",
]
SESSIONS_SYMBOLS = [
    "WebRtcAecmSessions_Create", "WebRtcAecmSessions_Free", "WebRtcAecmSessions_Init", "WebRtcAecmSessions_set_config",
    "WebRtcAecmSessions_Tick", "WebRtcAecmSessions_TickHost", "WebRtcAecmSessions_TickPerSession",
    "WebRtcAecmSessions_TickPerSessionHost", "WebRtcAecmSessions_TickFlags", "WebRtcAecmSessions_TickFlagsHost",
    "WebRtcAecmSessions_InitSession", "WebRtcAecmSessions_set_config_session", "WebRtcAecmSessions_InitEchoPath",
    "WebRtcAecmSessions_GetEchoPath", "WebRtcAecmSessions_TickAsync", "WebRtcAecmSessions_Synchronize",
    "WebRtcAecmSessions_SetKernelVariant", "WebRtcAecmSessions_session_size_bytes", "WebRtcAecmSessions_ExportSession",
    "WebRtcAecmSessions_ImportSession", "WebRtcAecmSessions_BufferFarend", "WebRtcAecmSessions_BufferFarendHost",
    "WebRtcAecmSessions_BufferFarendAsync", "WebRtcAecmSessions_Process", "WebRtcAecmSessions_ProcessHost", "WebRtcAecmSessions_DescribeTick",
]
SESSION_NO_FAREND = 1
SESSION_SPLIT_CALLS = 2


class AecmConfig(C.Structure):
    _fields_ = [("cngMode", C.c_int16), ("echoMode", C.c_int16)]


_lib = None


class AecmLaunchPolicy(C.Structure):
    """include/aecm_batch.h: AecmLaunchPolicy (every threshold and wish that decides how a ProcessBlocks launch is scheduled)."""
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "compute_units", "queue_chunk_blocks", "queue_chunk_explicit", "queue_min_streams", "pipelined_min_streams",
        "pipelined_min_blocks", "pipelined_max_streams", "resident_waves", "rotation_stream_limit", "pipe_tail_waves", "pipe_front_waves",
        "pipe_raw", "pipe_delay_waves", "pipe_gain_waves", "pipe_spread", "pipe_wgs_per_cu", "pipe_rot", "pipe_prio")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class AecmLaunchDescription(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("form", "chunk_blocks", "shape", "workgroups", "waves_per_workgroup", "workgroups_per_cu", "rounds_x1000", "cu_load_evenness_x1000")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def library_path() -> Path:
    """The library load() binds: $AECM_LIB_PATH (kernel A/B experiments) or the in-tree build."""
    return Path(os.environ["AECM_LIB_PATH"]) if os.environ.get("AECM_LIB_PATH") else _build.LIB


def load():
    """Load the HIP library (building it first if the sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64; two HIP runtimes in one process do not both see the
    # GPU.  Importing torch first makes this library bind to the runtime torch already loaded, so
    # torch tensors (device memory, streams) and this engine share one runtime.  Without torch
    # installed the system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    # AECM_LIB_PATH: load a specific prebuilt library (kernel A/B experiments) instead of (re)building.
    path = os.environ.get("AECM_LIB_PATH") or _build.build()
    lib = C.CDLL(str(path))
    vp, i16p = C.c_void_p, C.c_void_p
    lib.WebRtcAecm_Create.restype = vp
    lib.WebRtcAecm_Free.argtypes = [vp]
    lib.WebRtcAecm_Free.restype = None
    lib.WebRtcAecm_Init.argtypes = [vp, C.c_int32]
    lib.WebRtcAecm_BufferFarend.argtypes = [vp, i16p, C.c_size_t]
    lib.WebRtcAecm_GetBufferFarendError.argtypes = [vp, i16p, C.c_size_t]
    lib.WebRtcAecm_Process.argtypes = [vp, i16p, i16p, i16p, C.c_size_t, C.c_int16]
    lib.WebRtcAecm_set_config.argtypes = [vp, AecmConfig]
    lib.WebRtcAecm_InitEchoPath.argtypes = [vp, vp, C.c_size_t]
    lib.WebRtcAecm_GetEchoPath.argtypes = [vp, vp, C.c_size_t]
    lib.WebRtcAecm_echo_path_size_bytes.restype = C.c_size_t
    lib.WebRtcAecmBatch_Create.restype = vp
    lib.WebRtcAecmBatch_Create.argtypes = [C.c_int32, C.c_int32]
    lib.WebRtcAecmBatch_Free.argtypes = [vp]
    lib.WebRtcAecmBatch_Free.restype = None
    lib.WebRtcAecmBatch_num_streams.argtypes = [vp]
    lib.WebRtcAecmBatch_Init.argtypes = [vp, C.c_int32]
    lib.WebRtcAecmBatch_set_config.argtypes = [vp, AecmConfig, C.c_int32, C.c_int32]
    lib.WebRtcAecmBatch_Control.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.WebRtcAecmBatch_ProcessBlocks.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_int64, C.c_int32]
    lib.WebRtcAecmBatch_ProcessBlocksHost.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_int64, C.c_int32]
    lib.WebRtcAecmBatch_ProcessRecordings.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, C.c_int16]
    lib.WebRtcAecmBatch_ProcessRecordingsHost.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, C.c_int16]
    lib.WebRtcAecmBatch_Synchronize.argtypes = [vp]
    lib.WebRtcAecmBatch_GetLastLaunchMs.argtypes = [vp, C.POINTER(C.c_float)]
    lib.WebRtcAecmBatch_GetTimers.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.WebRtcAecmBatch_ResetTimers.argtypes = [vp]
    lib.WebRtcAecmBatch_InitEchoPath.argtypes = [vp, C.c_int32, vp, C.c_size_t]
    lib.WebRtcAecmBatch_GetEchoPath.argtypes = [vp, C.c_int32, vp, C.c_size_t]
    lib.WebRtcAecmBatch_state_size_bytes.restype = C.c_size_t
    lib.WebRtcAecmBatch_ExportState.argtypes = [vp, C.c_int32, vp, C.c_size_t]
    lib.WebRtcAecmBatch_ImportState.argtypes = [vp, C.c_int32, vp, C.c_size_t]
    lib.WebRtcAecmBatch_GetDigest.argtypes = [vp, C.c_int32, vp]
    for name in ("ExportStates", "ImportStates", "ExportStatesDevice", "ImportStatesDevice"):
        getattr(lib, "WebRtcAecmBatch_" + name).argtypes = [vp, C.c_int32, C.c_int32, vp, C.c_size_t]
    lib.WebRtcAecmSessions_session_size_bytes.restype = C.c_size_t
    lib.WebRtcAecmSessions_ExportSession.argtypes = [vp, C.c_int32, vp, C.c_size_t]
    lib.WebRtcAecmSessions_ImportSession.argtypes = [vp, C.c_int32, vp, C.c_size_t]
    lib.WebRtcAecmBatch_SetKernelVariant.argtypes = [vp, C.c_int32]
    lib.WebRtcAecmBatch_SetLaunchChunking.argtypes = [vp, C.c_int32, C.c_int32]
    lib.WebRtcAecmBatch_DescribeLaunch.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    lib.WebRtcAecmBatch_SetLaunchPipelining.argtypes = [vp, C.c_int32]
    lib.WebRtcAecmBatch_DescribeLaunchFor.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    lib.WebRtcAecmBatch_DefaultLaunchPolicy.argtypes = [C.c_int32, C.POINTER(AecmLaunchPolicy)]
    lib.WebRtcAecmBatch_GetLaunchPolicy.argtypes = [vp, C.POINTER(AecmLaunchPolicy)]
    lib.WebRtcAecmBatch_SetLaunchPolicy.argtypes = [vp, C.POINTER(AecmLaunchPolicy)]
    lib.WebRtcAecmBatch_DescribeLaunchDetail.argtypes = [C.POINTER(AecmLaunchPolicy), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                         C.POINTER(AecmLaunchDescription)]
    lib.WebRtcAecmSessions_DescribeTick.argtypes = [C.c_int32, C.c_int32, C.POINTER(AecmLaunchDescription)]
    lib.WebRtcAecm_SetDefaultDevice.argtypes = [C.c_int32]
    lib.WebRtcAecmSessions_Create.restype = vp
    lib.WebRtcAecmSessions_Create.argtypes = [C.c_int32, C.c_int32]
    lib.WebRtcAecmSessions_Free.argtypes = [vp]
    lib.WebRtcAecmSessions_Free.restype = None
    lib.WebRtcAecmSessions_Init.argtypes = [vp, C.c_int32]
    lib.WebRtcAecmSessions_set_config.argtypes = [vp, AecmConfig]
    lib.WebRtcAecmSessions_Tick.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_size_t, C.c_int16]
    lib.WebRtcAecmSessions_TickHost.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_size_t, C.c_int16]
    lib.WebRtcAecmSessions_TickPerSession.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_size_t, vp, vp]
    lib.WebRtcAecmSessions_TickPerSessionHost.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_size_t, vp, vp]
    lib.WebRtcAecmSessions_TickFlags.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_size_t, vp, vp, vp]
    lib.WebRtcAecmSessions_TickFlagsHost.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_size_t, vp, vp, vp]
    lib.WebRtcAecmSessions_InitSession.argtypes = [vp, C.c_int32]
    lib.WebRtcAecmSessions_set_config_session.argtypes = [vp, C.c_int32, AecmConfig]
    lib.WebRtcAecmSessions_InitEchoPath.argtypes = [vp, C.c_int32, vp, C.c_size_t]
    lib.WebRtcAecmSessions_GetEchoPath.argtypes = [vp, C.c_int32, vp, C.c_size_t]
    lib.WebRtcAecmSessions_TickAsync.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_size_t, C.c_int16, vp, vp, vp, vp, vp]
    lib.WebRtcAecmSessions_BufferFarend.argtypes = [vp, vp, C.c_int64, C.c_size_t, C.c_int32, vp]
    lib.WebRtcAecmSessions_BufferFarendHost.argtypes = [vp, vp, C.c_int64, C.c_size_t, C.c_int32, vp]
    lib.WebRtcAecmSessions_BufferFarendAsync.argtypes = [vp, vp, C.c_int64, C.c_size_t, C.c_int32, vp, vp, vp]
    lib.WebRtcAecmSessions_Process.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_size_t, C.c_int16, vp, vp]
    lib.WebRtcAecmSessions_ProcessHost.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_size_t, C.c_int16, vp, vp]
    lib.WebRtcAecmSessions_Synchronize.argtypes = [vp]
    lib.WebRtcAecmSessions_SetKernelVariant.argtypes = [vp, C.c_int32]
    lib.WebRtcAecmBatch_RegisterHostBuffer.argtypes = [C.c_int32, vp, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.WebRtcAecmBatch_UnregisterHostBuffer.argtypes = [C.c_int32, vp]
    lib.WebRtcAecmBatch_GetCheckCounters.argtypes = [C.c_int32, vp, C.c_int32]
    lib.WebRtcAecmBatch_SelfTest.argtypes = [C.c_int32, C.c_int32, vp]
    lib.WebRtcAecmBatch_DebugFft128.argtypes = [C.c_int32, vp, vp, C.c_int32, C.c_int32, C.c_int32]
    lib.WebRtcAecmBatch_DeviceInfo.argtypes = [C.c_int32, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.WebRtcAecmBatch_DevicePciBusId.argtypes = [C.c_int32, C.c_char_p, C.c_size_t]
    _lib = lib
    return lib


def _i16(a):
    a = np.ascontiguousarray(a, dtype=np.int16)
    return a, a.ctypes.data


class AecmError(RuntimeError):
    def __init__(self, code, where):
        super().__init__(f"{where} returned {code}")
        self.code = code


class Aecm:
    """One AECM session: the reference's WebRtcAecm_* interface, DSP on the GPU."""

    def __init__(self):
        self.lib = load()
        self.h = self.lib.WebRtcAecm_Create()
        if not self.h:
            raise RuntimeError("WebRtcAecm_Create failed: the HIP engine needs a usable MI355X (no CPU fallback exists)")

    def init(self, samp_freq: int) -> int:
        return self.lib.WebRtcAecm_Init(self.h, samp_freq)

    def set_config(self, cng_mode: int, echo_mode: int) -> int:
        return self.lib.WebRtcAecm_set_config(self.h, AecmConfig(cng_mode, echo_mode))

    def buffer_farend(self, farend) -> int:
        a, p = _i16(farend)
        return self.lib.WebRtcAecm_BufferFarend(self.h, p, a.size)

    def process(self, nearend_noisy, nearend_clean=None, ms_in_snd_card_buf: int = 0):
        """Returns (code, out)."""
        a, p = _i16(nearend_noisy)
        cp = None
        if nearend_clean is not None:
            c, cp = _i16(nearend_clean)
        out = np.empty_like(a)
        rc = self.lib.WebRtcAecm_Process(self.h, p, cp, out.ctypes.data, a.size, ms_in_snd_card_buf)
        return rc, out

    def init_echo_path(self, path) -> int:
        a, p = _i16(path)
        return self.lib.WebRtcAecm_InitEchoPath(self.h, p, a.nbytes)

    def get_echo_path(self):
        out = np.zeros(BINS, dtype=np.int16)
        rc = self.lib.WebRtcAecm_GetEchoPath(self.h, out.ctypes.data, out.nbytes)
        return rc, out

    def run(self, far, near, frame: int, ms: int = 40):
        """The reference CLI's loop (main.cc:105-143): BufferFarend + Process per `frame` samples."""
        far = np.ascontiguousarray(far, dtype=np.int16)
        near = np.ascontiguousarray(near, dtype=np.int16).copy()
        for i in range(near.size // frame):
            sl = slice(i * frame, (i + 1) * frame)
            rc = self.buffer_farend(far[sl])
            if rc != 0:
                raise AecmError(rc, "WebRtcAecm_BufferFarend")
            rc, out = self.process(near[sl], None, ms)
            if rc != 0:
                raise AecmError(rc, "WebRtcAecm_Process")
            near[sl] = out
        return near

    def close(self):
        if getattr(self, "h", None):
            self.lib.WebRtcAecm_Free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AecmBatch:
    """S independent AECM block streams on one GPU (include/aecm_batch.h)."""

    def __init__(self, num_streams: int, fs: int = 16000, cng_mode: int = 1, echo_mode: int = 3, device: int = 0,
                 variant: int = KERNEL_FAST):
        self.lib = load()
        self.num_streams = num_streams
        self.h = self.lib.WebRtcAecmBatch_Create(num_streams, device)
        if not self.h:
            raise RuntimeError("WebRtcAecmBatch_Create failed: the HIP engine needs a usable MI355X (no CPU fallback exists)")
        self._check(self.lib.WebRtcAecmBatch_Init(self.h, fs), "Init")
        self._check(self.lib.WebRtcAecmBatch_set_config(self.h, AecmConfig(cng_mode, echo_mode), 0, -1), "set_config")
        self._check(self.lib.WebRtcAecmBatch_SetKernelVariant(self.h, variant), "SetKernelVariant")

    @staticmethod
    def _check(rc, where):
        if rc != 0:
            raise AecmError(rc, "WebRtcAecmBatch_" + where)

    def set_launch_chunking(self, chunk_blocks, min_streams=-1):
        """Scheduling of large launches (results never depend on it): chunks of chunk_blocks blocks claimed from a queue
        by resident wavefronts; 0 = one wavefront per stream for the whole launch."""
        self._check(self.lib.WebRtcAecmBatch_SetLaunchChunking(self.h, chunk_blocks, min_streams), "SetLaunchChunking")

    def set_launch_pipelining(self, min_streams):
        """Smallest batch whose launches run pipelined (six to sixteen wavefronts per four streams; results never depend on it);
        <= 0: never.  By default launches of one or two blocks keep one wavefront per stream; after this call launches of
        any length do what min_streams says."""
        self._check(self.lib.WebRtcAecmBatch_SetLaunchPipelining(self.h, min_streams), "SetLaunchPipelining")

    def launch_policy(self) -> AecmLaunchPolicy:
        p = AecmLaunchPolicy()
        self._check(self.lib.WebRtcAecmBatch_GetLaunchPolicy(self.h, C.byref(p)), "WebRtcAecmBatch_GetLaunchPolicy")
        return p

    def set_launch_policy(self, policy=None, **fields):
        """Set the batch's launch policy: a whole AecmLaunchPolicy, or the current one with `fields` changed
        (e.g. pipe_tail_waves=0, queue_min_streams=0)."""
        p = policy if policy is not None else self.launch_policy()
        for k, v in fields.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        self._check(self.lib.WebRtcAecmBatch_SetLaunchPolicy(self.h, C.byref(p)), "WebRtcAecmBatch_SetLaunchPolicy")

    def describe_launch(self, num_blocks, clean=False):
        """(form, chunk_blocks) of a ProcessBlocks launch of num_blocks blocks: form 0 / 1 = one wavefront per stream
        (small-launch variants / issue priority by phase), 2 = chunk queue, 3 = pipelined (chunk_blocks is then the shape: include/aecm_batch.h)."""
        chunk = C.c_int32(0)
        form = self.lib.WebRtcAecmBatch_DescribeLaunch(self.h, num_blocks, 1 if clean else 0, C.byref(chunk))
        if form < 0:
            raise AecmError(form, "WebRtcAecmBatch_DescribeLaunch")
        return form, chunk.value

    def set_config(self, cng_mode, echo_mode, first=0, count=-1):
        self._check(self.lib.WebRtcAecmBatch_set_config(self.h, AecmConfig(cng_mode, echo_mode), first, count), "set_config")

    def control(self, fixed_delay, nlp_flag, first=0, count=-1):
        self._check(self.lib.WebRtcAecmBatch_Control(self.h, fixed_delay, nlp_flag, first, count), "Control")

    def set_variant(self, variant):
        self._check(self.lib.WebRtcAecmBatch_SetKernelVariant(self.h, variant), "SetKernelVariant")

    def process_host(self, far, near, clean=None, out=None):
        """far/near: [S, T*64] int16 host arrays (stream-major).  Returns out [S, T*64] (written into `out` if given)."""
        far = np.ascontiguousarray(far, dtype=np.int16)
        near = np.ascontiguousarray(near, dtype=np.int16)
        assert far.shape == near.shape and far.shape[0] == self.num_streams and far.shape[1] % BLOCK == 0
        if out is None:
            out = np.empty_like(near)
        assert out.shape == near.shape and out.dtype == np.int16 and out.flags.c_contiguous
        cp = None
        if clean is not None:
            clean = np.ascontiguousarray(clean, dtype=np.int16)
            assert clean.shape == near.shape, "clean must have the shape of near"
            cp = clean.ctypes.data
        t = far.shape[1] // BLOCK
        self._check(self.lib.WebRtcAecmBatch_ProcessBlocksHost(self.h, far.ctypes.data, near.ctypes.data, cp,
                                                               out.ctypes.data, far.shape[1], BLOCK, t), "ProcessBlocksHost")
        return out

    def process_recordings_host(self, far, near, frame: int, ms: int = 40, clean=None):
        """far/near(/clean): [S, N] int16 host arrays, each stream a whole recording driven like the reference
        CLI (BufferFarend + Process per `frame` samples, constant msInSndCardBuf; clean = nearendClean).
        Returns (code, out)."""
        far = np.ascontiguousarray(far, dtype=np.int16)
        near = np.ascontiguousarray(near, dtype=np.int16)
        assert far.shape == near.shape and far.shape[0] == self.num_streams
        cptr = None
        if clean is not None:
            clean = np.ascontiguousarray(clean, dtype=np.int16)
            assert clean.shape == near.shape
            cptr = clean.ctypes.data
        out = near.copy()
        n_calls = far.shape[1] // frame
        rc = self.lib.WebRtcAecmBatch_ProcessRecordingsHost(self.h, far.ctypes.data, near.ctypes.data, cptr, out.ctypes.data,
                                                            far.shape[1], frame, n_calls, ms)
        return rc, out

    def process_recordings_device(self, far_ptr, near_ptr, out_ptr, stream_stride, frame, n_calls, ms=40, clean_ptr=None):
        return self.lib.WebRtcAecmBatch_ProcessRecordings(self.h, far_ptr, near_ptr, clean_ptr, out_ptr, stream_stride, frame,
                                                          n_calls, ms)

    def process_device(self, far_ptr, near_ptr, out_ptr, stream_stride, block_stride, num_blocks, clean_ptr=None):
        """Device pointers (e.g. torch .data_ptr()); asynchronous on the engine's stream."""
        self._check(self.lib.WebRtcAecmBatch_ProcessBlocks(self.h, far_ptr, near_ptr, clean_ptr, out_ptr, stream_stride,
                                                           block_stride, num_blocks), "ProcessBlocks")

    def synchronize(self):
        self._check(self.lib.WebRtcAecmBatch_Synchronize(self.h), "Synchronize")

    def last_launch_ms(self) -> float:
        ms = C.c_float()
        self._check(self.lib.WebRtcAecmBatch_GetLastLaunchMs(self.h, C.byref(ms)), "GetLastLaunchMs")
        return ms.value

    def timers(self):
        total, n = C.c_double(), C.c_int64()
        self._check(self.lib.WebRtcAecmBatch_GetTimers(self.h, C.byref(total), C.byref(n)), "GetTimers")
        return total.value, n.value

    def reset_timers(self):
        self._check(self.lib.WebRtcAecmBatch_ResetTimers(self.h), "ResetTimers")

    def digest(self, stream: int):
        d = np.zeros(DIGEST_WORDS, dtype=np.uint32)
        self._check(self.lib.WebRtcAecmBatch_GetDigest(self.h, stream, d.ctypes.data), "GetDigest")
        return d

    def export_state(self, stream: int) -> bytes:
        buf = C.create_string_buffer(self.lib.WebRtcAecmBatch_state_size_bytes())
        self._check(self.lib.WebRtcAecmBatch_ExportState(self.h, stream, buf, len(buf)), "ExportState")
        return buf.raw

    def import_state(self, stream: int, state: bytes):
        self._check(self.lib.WebRtcAecmBatch_ImportState(self.h, stream, state, len(state)), "ImportState")

    def export_states(self, first: int, count: int) -> np.ndarray:
        """Snapshots of streams [first, first + count) as a [count, state_size] uint8 array (one gather launch + chunked copies)."""
        n = self.lib.WebRtcAecmBatch_state_size_bytes()
        buf = np.empty((count, n), dtype=np.uint8)
        self._check(self.lib.WebRtcAecmBatch_ExportStates(self.h, first, count, buf.ctypes.data, buf.nbytes), "ExportStates")
        return buf

    def import_states(self, first: int, states: np.ndarray):
        states = np.ascontiguousarray(states, dtype=np.uint8)
        self._check(self.lib.WebRtcAecmBatch_ImportStates(self.h, first, states.shape[0], states.ctypes.data, states.nbytes), "ImportStates")

    def export_states_device(self, first: int, count: int, dev_ptr: int):
        n = self.lib.WebRtcAecmBatch_state_size_bytes()
        self._check(self.lib.WebRtcAecmBatch_ExportStatesDevice(self.h, first, count, dev_ptr, count * n), "ExportStatesDevice")

    def import_states_device(self, first: int, count: int, dev_ptr: int):
        n = self.lib.WebRtcAecmBatch_state_size_bytes()
        self._check(self.lib.WebRtcAecmBatch_ImportStatesDevice(self.h, first, count, dev_ptr, count * n), "ImportStatesDevice")

    def init_echo_path(self, stream, path):
        a, p = _i16(path)
        self._check(self.lib.WebRtcAecmBatch_InitEchoPath(self.h, stream, p, a.nbytes), "InitEchoPath")

    def get_echo_path(self, stream):
        out = np.zeros(BINS, dtype=np.int16)
        self._check(self.lib.WebRtcAecmBatch_GetEchoPath(self.h, stream, out.ctypes.data, out.nbytes), "GetEchoPath")
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.WebRtcAecmBatch_Free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AecmSessions:
    """S streaming sessions with a common call cadence (include/aecm_batch.h, WebRtcAecmSessions_*)."""

    def __init__(self, num_streams: int, fs: int = 16000, cng_mode: int = 1, echo_mode: int = 3, device: int = 0):
        self.lib = load()
        self.num_streams = num_streams
        self.h = self.lib.WebRtcAecmSessions_Create(num_streams, device)
        if not self.h:
            raise RuntimeError("WebRtcAecmSessions_Create failed: the HIP engine needs a usable MI355X (no CPU fallback exists)")
        rc = self.lib.WebRtcAecmSessions_Init(self.h, fs)
        if rc != 0:
            raise AecmError(rc, "WebRtcAecmSessions_Init")
        rc = self.lib.WebRtcAecmSessions_set_config(self.h, AecmConfig(cng_mode, echo_mode))
        if rc != 0:
            raise AecmError(rc, "WebRtcAecmSessions_set_config")

    def _rows(self, far, near, clean):
        """Validated contiguous [S, n] int16 rows (the C side strides by far.shape[1] and reads S rows)."""
        far = np.ascontiguousarray(far, dtype=np.int16)
        near = np.ascontiguousarray(near, dtype=np.int16)
        if far.ndim != 2 or far.shape[0] != self.num_streams or far.shape[1] not in (80, 160):
            raise ValueError(f"far must be [{self.num_streams}, 80 or 160] int16, got {far.shape}")
        if near.shape != far.shape:
            raise ValueError(f"near {near.shape} must have the shape of far {far.shape}")
        cptr = None
        if clean is not None:
            clean = np.ascontiguousarray(clean, dtype=np.int16)
            if clean.shape != far.shape:
                raise ValueError(f"clean {clean.shape} must have the shape of far {far.shape}")
            cptr = clean.ctypes.data
        return far, near, clean, cptr

    def tick_host(self, far, near, ms: int = 40, clean=None):
        """far/near(/clean): [S, n] int16 (n = 80 or 160).  Returns (code, out)."""
        far, near, clean, cptr = self._rows(far, near, clean)
        out = np.empty_like(near)
        rc = self.lib.WebRtcAecmSessions_TickHost(self.h, far.ctypes.data, near.ctypes.data, cptr, out.ctypes.data, far.shape[1],
                                                  far.shape[1], ms)
        return rc, out

    def tick_device(self, far_ptr, near_ptr, out_ptr, stream_stride, n, ms=40, clean_ptr=None):
        return self.lib.WebRtcAecmSessions_Tick(self.h, far_ptr, near_ptr, clean_ptr, out_ptr, stream_stride, n, ms)

    def tick_host_per_session(self, far, near, ms_per_session, clean=None, flags=None):
        """far/near(/clean): [S, n] int16; ms_per_session: S msInSndCardBuf values; flags: S uint8 (SESSION_NO_FAREND)
        or None.  Returns (code, out, codes[S])."""
        far, near, clean, cptr = self._rows(far, near, clean)
        ms = np.ascontiguousarray(ms_per_session, dtype=np.int16)
        if ms.shape != (self.num_streams,):
            raise ValueError("ms_per_session must have one entry per session")
        out = np.empty_like(near)
        codes = np.zeros(far.shape[0], dtype=np.int32)
        if flags is None:
            rc = self.lib.WebRtcAecmSessions_TickPerSessionHost(self.h, far.ctypes.data, near.ctypes.data, cptr, out.ctypes.data,
                                                                far.shape[1], far.shape[1], ms.ctypes.data, codes.ctypes.data)
        else:
            fl = np.ascontiguousarray(flags, dtype=np.uint8)
            if fl.shape != (self.num_streams,):
                raise ValueError("flags must have one entry per session")
            rc = self.lib.WebRtcAecmSessions_TickFlagsHost(self.h, far.ctypes.data, near.ctypes.data, cptr, out.ctypes.data,
                                                           far.shape[1], far.shape[1], ms.ctypes.data, fl.ctypes.data,
                                                           codes.ctypes.data)
        return rc, out, codes

    def buffer_farend_host(self, far, n: int, calls: int, calls_per_session=None) -> int:
        """Far-end burst (include/aecm_batch.h: WebRtcAecmSessions_BufferFarendHost): far [S, >= calls * n] int16; session s
        makes calls_per_session[s] (or `calls`) WebRtcAecm_BufferFarend calls of n samples."""
        far = np.ascontiguousarray(far, dtype=np.int16)
        if far.ndim != 2 or far.shape[0] != self.num_streams:
            raise ValueError(f"far must be [{self.num_streams}, >= calls * n] int16, got {far.shape}")
        cp = None
        if calls_per_session is not None:
            calls_per_session = np.ascontiguousarray(calls_per_session, dtype=np.uint8)
            if calls_per_session.shape != (self.num_streams,):
                raise ValueError("calls_per_session must have one entry per session")
            cp = calls_per_session.ctypes.data
        return self.lib.WebRtcAecmSessions_BufferFarendHost(self.h, far.ctypes.data, far.shape[1], n, calls, cp)

    def buffer_farend_device(self, far_ptr, stream_stride, n, calls, calls_per_session=None, asynchronous=False, wait_event=None, done_event=None):
        cp = None
        if calls_per_session is not None:
            calls_per_session = np.ascontiguousarray(calls_per_session, dtype=np.uint8)
            cp = calls_per_session.ctypes.data
        if asynchronous:
            return self.lib.WebRtcAecmSessions_BufferFarendAsync(self.h, far_ptr, stream_stride, n, calls, cp, wait_event, done_event)
        return self.lib.WebRtcAecmSessions_BufferFarend(self.h, far_ptr, stream_stride, n, calls, cp)

    def process_host(self, near, ms=40, clean=None, ms_per_session=None):
        """Every session's WebRtcAecm_Process without a WebRtcAecm_BufferFarend (WebRtcAecmSessions_ProcessHost).
        Returns (code, out, codes[S])."""
        near = np.ascontiguousarray(near, dtype=np.int16)
        if near.ndim != 2 or near.shape[0] != self.num_streams:
            raise ValueError(f"near must be [{self.num_streams}, n] int16, got {near.shape}")
        cptr = msp = None
        if clean is not None:
            clean = np.ascontiguousarray(clean, dtype=np.int16)
            if clean.shape != near.shape:
                raise ValueError("clean must have the shape of near")
            cptr = clean.ctypes.data
        if ms_per_session is not None:
            ms_per_session = np.ascontiguousarray(ms_per_session, dtype=np.int16)
            if ms_per_session.shape != (self.num_streams,):
                raise ValueError("ms_per_session must have one entry per session")
            msp = ms_per_session.ctypes.data
        out = np.empty_like(near)
        codes = np.zeros(self.num_streams, dtype=np.int32)
        rc = self.lib.WebRtcAecmSessions_ProcessHost(self.h, near.ctypes.data, cptr, out.ctypes.data, near.shape[1], near.shape[1], ms, msp,
                                                     codes.ctypes.data)
        return rc, out, codes

    def process_device(self, near_ptr, out_ptr, stream_stride, n, ms=40, clean_ptr=None, ms_per_session=None):
        msp = None
        if ms_per_session is not None:
            ms_per_session = np.ascontiguousarray(ms_per_session, dtype=np.int16)
            msp = ms_per_session.ctypes.data
        return self.lib.WebRtcAecmSessions_Process(self.h, near_ptr, clean_ptr, out_ptr, stream_stride, n, ms, msp, None)

    def init_session(self, session: int) -> int:
        return self.lib.WebRtcAecmSessions_InitSession(self.h, session)

    def set_config_session(self, session: int, cng_mode: int, echo_mode: int) -> int:
        return self.lib.WebRtcAecmSessions_set_config_session(self.h, session, AecmConfig(cng_mode, echo_mode))

    def init_echo_path(self, session: int, path) -> int:
        a, p = _i16(path)
        return self.lib.WebRtcAecmSessions_InitEchoPath(self.h, session, p, a.nbytes)

    def get_echo_path(self, session: int):
        out = np.zeros(BINS, dtype=np.int16)
        rc = self.lib.WebRtcAecmSessions_GetEchoPath(self.h, session, out.ctypes.data, out.nbytes)
        return rc, out

    def export_session(self, session: int):
        """(code, snapshot bytes) of one live session (include/aecm_batch.h: WebRtcAecmSessions_ExportSession)."""
        buf = C.create_string_buffer(self.lib.WebRtcAecmSessions_session_size_bytes())
        rc = self.lib.WebRtcAecmSessions_ExportSession(self.h, session, buf, len(buf))
        return rc, buf.raw

    def import_session(self, session: int, snapshot: bytes) -> int:
        return self.lib.WebRtcAecmSessions_ImportSession(self.h, session, snapshot, len(snapshot))

    def tick_device_per_session(self, far_ptr, near_ptr, out_ptr, stream_stride, n, ms_per_session, clean_ptr=None):
        ms = np.ascontiguousarray(ms_per_session, dtype=np.int16)
        if ms.shape != (self.num_streams,):
            raise ValueError("ms_per_session must have one entry per session")
        return self.lib.WebRtcAecmSessions_TickPerSession(self.h, far_ptr, near_ptr, clean_ptr, out_ptr, stream_stride, n,
                                                          ms.ctypes.data, None)

    def tick_async(self, far_ptr, near_ptr, out_ptr, stream_stride, n, ms=40, clean_ptr=None, ms_per_session=None, flags=None,
                   wait_event=None, done_event=None):
        """Enqueue one tick and return without waiting for it (device pointers); see include/aecm_batch.h."""
        msp = flp = None
        if ms_per_session is not None:
            ms_per_session = np.ascontiguousarray(ms_per_session, dtype=np.int16)
            msp = ms_per_session.ctypes.data
        if flags is not None:
            flags = np.ascontiguousarray(flags, dtype=np.uint8)
            flp = flags.ctypes.data
        return self.lib.WebRtcAecmSessions_TickAsync(self.h, far_ptr, near_ptr, clean_ptr, out_ptr, stream_stride, n, ms, msp, flp, None,
                                                     wait_event, done_event)

    def synchronize(self):
        return self.lib.WebRtcAecmSessions_Synchronize(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.WebRtcAecmSessions_Free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def register_host_buffer(array, device: int = 0) -> int:
    """Pin + map a caller-owned numpy array (zero-copy host audio); returns its device alias for the *_dev arguments."""
    lib = load()
    dev = C.c_void_p()
    rc = lib.WebRtcAecmBatch_RegisterHostBuffer(device, array.ctypes.data, array.nbytes, C.byref(dev))
    if rc != 0:
        raise AecmError(rc, "WebRtcAecmBatch_RegisterHostBuffer")
    return dev.value


def unregister_host_buffer(array, device: int = 0) -> None:
    rc = load().WebRtcAecmBatch_UnregisterHostBuffer(device, array.ctypes.data)
    if rc != 0:
        raise AecmError(rc, "WebRtcAecmBatch_UnregisterHostBuffer")


def self_test(device: int = 0, exhaustive: bool = False):
    """Device self test of the wave primitives; returns the 8 failure counters (all must be 0)."""
    lib = load()
    f = np.zeros(8, dtype=np.uint64)
    rc = lib.WebRtcAecmBatch_SelfTest(device, 1 if exhaustive else 0, f.ctypes.data)
    if rc != 0:
        raise AecmError(rc, "WebRtcAecmBatch_SelfTest")
    return f


def check_counters(device: int = 0, reset: bool = False):
    """Precondition-violation counters of the audit build (AECM_LIB_PATH=.../libaecm_mi355x_checked.so):
    [mul24 operands, as_i16 arguments].  Raises AecmError(12001) on the shipped library."""
    lib = load()
    c = np.zeros(2, dtype=np.uint64)
    rc = lib.WebRtcAecmBatch_GetCheckCounters(device, c.ctypes.data, 1 if reset else 0)
    if rc != 0:
        raise AecmError(rc, "WebRtcAecmBatch_GetCheckCounters")
    return c


def debug_fft128(re, im, variant: int, kernel_variant: int = KERNEL_FAST, device: int = 0):
    """Run the block kernel's own 128-point transform on (count, 128) int16 arrays; returns (re, im, scales).
    variant: 0 forward of real input, 1 forward complex, 2 inverse (see include/aecm_batch.h)."""
    lib = load()
    re = np.ascontiguousarray(re, dtype=np.int16).reshape(-1, 128)
    im = np.ascontiguousarray(im, dtype=np.int16).reshape(-1, 128)
    data = np.ascontiguousarray(np.concatenate([re, im], axis=1))
    scales = np.zeros(re.shape[0], dtype=np.int32)
    rc = lib.WebRtcAecmBatch_DebugFft128(device, data.ctypes.data, scales.ctypes.data, variant, kernel_variant, re.shape[0])
    if rc != 0:
        raise AecmError(rc, "WebRtcAecmBatch_DebugFft128")
    return data[:, :128].copy(), data[:, 128:].copy(), scales


def describe_launch_for(num_streams: int, compute_units: int, num_blocks: int, clean: bool = False):
    """(form, chunk_blocks / shape) a default-configured batch of num_streams streams would launch num_blocks blocks with on a
    device of compute_units CUs (AecmBatch.describe_launch without a device: the launch-form rules are host logic)."""
    chunk = C.c_int32(0)
    form = load().WebRtcAecmBatch_DescribeLaunchFor(num_streams, compute_units, num_blocks, 1 if clean else 0, C.byref(chunk))
    if form < 0:
        raise AecmError(form, "WebRtcAecmBatch_DescribeLaunchFor")
    return form, chunk.value


def default_launch_policy(compute_units: int) -> AecmLaunchPolicy:
    p = AecmLaunchPolicy()
    rc = load().WebRtcAecmBatch_DefaultLaunchPolicy(compute_units, C.byref(p))
    if rc != 0:
        raise AecmError(rc, "WebRtcAecmBatch_DefaultLaunchPolicy")
    return p


def describe_launch_detail(num_streams: int, num_blocks: int, compute_units: int = 0, clean: bool = False, policy=None) -> dict:
    """Form, shape, grid and chip quantisation of a launch (include/aecm_batch.h: AecmLaunchDescription), without a device:
    under `policy` (an AecmLaunchPolicy) or the default policy of compute_units."""
    d = AecmLaunchDescription()
    rc = load().WebRtcAecmBatch_DescribeLaunchDetail(C.byref(policy) if policy is not None else None, compute_units, num_streams, num_blocks,
                                                     1 if clean else 0, C.byref(d))
    if rc != 0:
        raise AecmError(rc, "WebRtcAecmBatch_DescribeLaunchDetail")
    return d.as_dict()


def describe_tick(num_sessions: int, compute_units: int) -> dict:
    d = AecmLaunchDescription()
    rc = load().WebRtcAecmSessions_DescribeTick(num_sessions, compute_units, C.byref(d))
    if rc != 0:
        raise AecmError(rc, "WebRtcAecmSessions_DescribeTick")
    return d.as_dict()


def set_default_device(device: int) -> int:
    return load().WebRtcAecm_SetDefaultDevice(device)


def device_pci_bus_id(device: int = 0) -> str:
    buf = C.create_string_buffer(32)
    rc = load().WebRtcAecmBatch_DevicePciBusId(device, buf, 32)
    if rc != 0:
        raise AecmError(rc, "WebRtcAecmBatch_DevicePciBusId")
    return buf.value.decode()


def device_info(device: int = 0):
    lib = load()
    name = C.create_string_buffer(64)
    cu, clk = C.c_int32(), C.c_int32()
    rc = lib.WebRtcAecmBatch_DeviceInfo(device, name, 64, C.byref(cu), C.byref(clk))
    if rc != 0:
        raise AecmError(rc, "WebRtcAecmBatch_DeviceInfo")
    return name.value.decode(), cu.value, clk.value
