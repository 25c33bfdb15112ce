"""HIP streams that belong to this package alone.

`torch.cuda.Stream(device)` hands out one of 32 pooled streams per device and priority, round robin: two "different"
Stream objects are the SAME hipStream_t after 32 allocations, and torch's own components (ProcessGroupNCCL among them)
draw from the same pool.  A stream that joins a hipGraph capture must not be one somebody else has recorded events on:
`hipEventQuery` of such an event from another thread (the process group's watchdog polls its Work events on its own
schedule) answers hipErrorCapturedEvent while the stream is capturing, and the watchdog turns that into std::terminate --
rounds 4-5 saw it in about one of five full GPU test runs and papered over it with a sleep.  Every stream the iteration
captures on -- the capture stream, the eager warm-up stream, the solver's second stream -- is therefore created here with
hipStreamCreateWithFlags and wrapped in `torch.cuda.ExternalStream`: never aliased, never seen by anybody else.
(No reference counterpart: the reference runs on torch's default stream, main.py:67-70.)
"""
import ctypes as C
import os

import torch

_hip = None
_streams = {}
HIP_STREAM_NON_BLOCKING = 1


def _runtime():
    """The HIP runtime already mapped into this process (torch's), found through /proc/self/maps so that no second copy of
    libamdhip64 is ever loaded."""
    global _hip
    if _hip is None:
        torch.cuda.init()
        path = None
        try:
            with open("/proc/self/maps") as f:
                for line in f:
                    if "libamdhip64" in line:
                        path = line.split()[-1]
                        break
        except OSError:
            pass
        _hip = C.CDLL(path or "libamdhip64.so")
        _hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        _hip.hipStreamCreateWithFlags.restype = C.c_int
    return _hip


def private_stream(device, tag="side"):
    """The package's own stream `tag` on `device` (created once per process; lives as long as the process)."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, tag)
    s = _streams.get(key)
    if s is None:
        handle = C.c_void_p()
        with torch.cuda.device(idx):
            err = _runtime().hipStreamCreateWithFlags(C.byref(handle), HIP_STREAM_NON_BLOCKING)
        if err != 0 or not handle.value:
            raise RuntimeError(f"hipStreamCreateWithFlags failed with hipError_t {err}")
        s = _streams[key] = torch.cuda.ExternalStream(handle.value, device=torch.device("cuda", idx))
    return s
