"""roctx ranges around the phases of train_step (tracing row: the reference has no tracing; the build supplies
rocprofv3 counters + roctx ranges per phase).  The ranges are host-side markers from librocprofiler-sdk-roctx: they cost about a
microsecond each, show up in `rocprofv3 --marker-trace --kernel-trace` timelines of EAGER steps (SDT_GRAPH=0: a replayed HIP graph
has no host-side phases) and are silently absent when the library is (it is a profiling aid, not part of the arithmetic)."""
import ctypes
import os
from contextlib import contextmanager

_lib = None
_tried = False
_SYNC = os.environ.get("SDT_ROCTX_SYNC") == "1"  # tools/refresh_profiles.sh marker pass: synchronise at range edges


def _load():
    global _lib, _tried
    if not _tried:
        _tried = True
        if os.environ.get("SDT_ROCTX", "1") != "0":
            for name in ("librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so"):
                try:
                    lib = ctypes.CDLL(name)
                    lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    lib.roctxRangePushA.restype = ctypes.c_int
                    lib.roctxRangePop.restype = ctypes.c_int
                    _lib = lib
                    break
                except (OSError, AttributeError):
                    continue
    return _lib


@contextmanager
def phase(name):
    """with trace.phase("unet_forward"): ...  (a roctx range when the marker library is present, nothing otherwise)"""
    lib = _load()
    if lib is None:
        yield
        return
    if _SYNC:  # profiling aid: make the host-side range bracket the device work of the phase (eager launches run ahead of the GPU)
        import torch
        torch.cuda.synchronize()
    lib.roctxRangePushA(name.encode())
    try:
        yield
        if _SYNC:
            torch.cuda.synchronize()
    finally:
        lib.roctxRangePop()
