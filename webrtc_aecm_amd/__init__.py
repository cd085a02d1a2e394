"""MI355X-native batched AECM (WebRTC acoustic echo canceller, mobile) block engine.

The hot path -- WebRtcAecm_ProcessBlock of cpuimage/WebRTC_AECM -- is hand-written HIP for gfx950
(one wavefront per stream, webrtc_aecm_amd/csrc/), exposed through the reference's own C ABI plus a
batch extension (include/*.h).  This package only builds and binds that shared library.
"""
from .ffi import (check_counters, Aecm, AecmBatch, AecmSessions, AecmConfig, AecmError, AecmLaunchPolicy, KERNEL_FAST, KERNEL_SAFE, debug_fft128,  # noqa: F401
                  default_launch_policy, describe_launch_detail, describe_launch_for, describe_tick, device_info, device_pci_bus_id, library_path, load,
                  register_host_buffer, self_test, set_default_device, unregister_host_buffer)

__all__ = ["check_counters", "Aecm", "AecmBatch", "AecmSessions", "AecmConfig", "AecmError", "AecmLaunchPolicy", "KERNEL_FAST", "KERNEL_SAFE", "debug_fft128",
           "default_launch_policy", "describe_launch_detail", "describe_launch_for", "describe_tick", "device_info", "device_pci_bus_id", "library_path", "load",
           "register_host_buffer", "self_test", "set_default_device", "unregister_host_buffer"]
